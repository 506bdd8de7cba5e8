#!/bin/bash
# rocprofv3 --kernel-trace of bench.py under a mode, then the step timeline (tools/trace_gaps.py).  usage: tools/prof_trace.sh <tag> [bench args]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=${1:-run}; shift || true
out=$R/gpurun_out/trace_$tag
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d "$out" -- python3 "$R/bench.py" --steps 100 --warmup 10 --no-cpu --no-events --no-secondary "$@" > "$out/bench.json" 2> "$out/bench.err"
echo "== $tag: rocprofv3 exit $?"
f=$(ls -t "$out"/*/*kernel_trace.csv 2>/dev/null | head -1)
python3 "$R/tools/trace_gaps.py" "$f"
