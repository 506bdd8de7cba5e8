#!/bin/bash
# LAB: bench.py under alternative builds of liblpf.so (lab_build/*.so) and environment settings.
# usage: tools/lab_so_sweep.sh "<so name or ->|<env assignments>|<bench args>" ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cp $R/lidar_object_detection_amd/liblpf.so /tmp/liblpf_orig.so
cd /tmp
for spec in "$@"; do
  IFS='|' read -r so envs bargs <<< "$spec"
  if [ "$so" != "-" ]; then cp $R/lab_build/$so $R/lidar_object_detection_amd/liblpf.so; else cp /tmp/liblpf_orig.so $R/lidar_object_detection_amd/liblpf.so; fi
  env $envs timeout -k 10 200 python3 $R/bench.py --no-cpu --no-secondary --steps 200 --warmup 20 $bargs > /tmp/o.json 2>/tmp/o.err || tail -3 /tmp/o.err
  python3 -c "
import json;d=json.loads(open('/tmp/o.json').read().strip().splitlines()[-1]);print('%-60s step %.2f us  bracket %.2f us' % ('$spec', 1e3*d['ms_per_step'], d['roofline']['avg_us']))"
done
cp /tmp/liblpf_orig.so $R/lidar_object_detection_amd/liblpf.so
