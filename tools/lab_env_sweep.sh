#!/bin/bash
# LAB: bench.py --mode fused under a list of environment settings, one line each.  usage: tools/lab_env_sweep.sh "A=1 B=2" "A=2" ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp
for e in "$@"; do
  env $e timeout -k 10 200 python3 $R/bench.py --no-cpu --no-secondary --steps 200 --warmup 20 --mode ${MODE:-fused} > /tmp/o.json 2>/tmp/o.err || tail -3 /tmp/o.err
  python3 -c "
import json;d=json.loads(open('/tmp/o.json').read().strip().splitlines()[-1]);print('%-40s step %.2f us  bracket %.2f us' % ('$e', 1e3*d['ms_per_step'], d['roofline']['avg_us']))"
done
