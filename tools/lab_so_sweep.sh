#!/bin/bash
# LAB: bench.py under alternative builds of liblpf.so, one line each -- same box, back to back (box-to-box spread is larger than most
# effects worth measuring).  An alternative build is made with tools/build_alt.py into lidar_object_detection_amd/build/alt/ (git-
# ignored; delete it when done: it travels to the GPU box) and chosen through LPF_LIBRARY, which bench.py honours in --lab runs only:
# the lines are marked "LAB RUN" and skip the oracle check -- never a reported number.  The package's own liblpf.so is never touched.
# usage: tools/lab_so_sweep.sh "<so name under build/alt/, or - for the product>|<bench args>" ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp
for spec in "$@"; do
  IFS='|' read -r so bargs <<< "$spec"
  lib=$R/lidar_object_detection_amd/liblpf.so
  [ "$so" != "-" ] && lib=$R/lidar_object_detection_amd/build/alt/$so
  LPF_LIBRARY=$lib timeout -k 10 200 python3 $R/bench.py --lab ab --no-cpu --no-secondary --steps 200 --warmup 20 $bargs > /tmp/o.json 2>/tmp/o.err || tail -3 /tmp/o.err
  python3 -c "
import json;d=json.loads(open('/tmp/o.json').read().strip().splitlines()[-1]);print('%-60s step %.2f us  bracket %.2f us' % ('$spec', 1e3*d['ms_per_step'], d['roofline']['avg_us']))"
done
