#!/bin/bash
# LAB: bench.py under alternative builds of liblpf.so (lab_build/*.so), one line each.  The library is chosen through the
# LPF_LIBRARY environment variable -- the package's own liblpf.so is never touched.
# usage: tools/lab_so_sweep.sh "<so name under lab_build/, or - for the product>|<bench args>" ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp
for spec in "$@"; do
  IFS='|' read -r so bargs <<< "$spec"
  lib=$R/lidar_object_detection_amd/liblpf.so
  [ "$so" != "-" ] && lib=$R/lab_build/$so
  LPF_LIBRARY=$lib timeout -k 10 200 python3 $R/bench.py --no-cpu --no-secondary --steps 200 --warmup 20 $bargs > /tmp/o.json 2>/tmp/o.err || tail -3 /tmp/o.err
  python3 -c "
import json;d=json.loads(open('/tmp/o.json').read().strip().splitlines()[-1]);print('%-60s step %.2f us  bracket %.2f us' % ('$spec', 1e3*d['ms_per_step'], d['roofline']['avg_us']))"
done
