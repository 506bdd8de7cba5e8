#!/bin/bash
# Development sweep (GPU box): the bench step under each queueing mode / CU partition, kernel averages by rocprofv3.
# usage: tools/mode_sweep.sh [quick]     outputs: gpurun_out/sweep/
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/sweep
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
run() {  # tag, bench args...
  local tag=$1; shift
  timeout -k 10 200 python3 "$R/bench.py" --no-cpu --no-secondary --steps 200 --warmup 20 "$@" > "$O/$tag.json" 2> "$O/$tag.err" || { echo "$tag FAILED"; tail -3 "$O/$tag.err"; return 1; }
  python3 - "$O/$tag.json" "$tag" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d.get("roofline",{})
print("%-28s %7.2f us/step  K1 bracket %6.2f us  (step frac %.3f)" % (sys.argv[2], 1e3*d["ms_per_step"], r.get("avg_us",0), d["config"]["step_algorithmic_frac_of_hbm_peak"]))
PY
}
if [ "${1:-}" = "quick" ]; then
run serial --mode serial && run fused --mode fused && run fused_pack --mode fused-pack && run pipeline --mode pipeline
else
run serial --mode serial && run fused --mode fused && run fused_pack --mode fused-pack && run pipeline --mode pipeline && run pipeline_pack --mode pipeline-pack && \
run part32 --mode partition --side-cus 32 && run part64 --mode partition --side-cus 64 && run part64x --mode partition --side-cus 64 --exclusive
fi
