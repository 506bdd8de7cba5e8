#!/bin/bash
# Kernel-time profile of bench.py on the GPU box: rocprofv3 --kernel-trace --stats (CSV), as used for
# profiles/r01_bench_kernel_stats.csv.  usage: tools/prof_bench.sh <tag> [bench.py args...]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=${1:-run}; shift || true
out=$R/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- \
    python3 "$R/bench.py" --steps 100 --warmup 10 --no-cpu --no-events --no-secondary "$@" > "$out/bench.json" 2> "$out/bench.err"
echo "rocprofv3 exit $?" >> "$out/bench.err"
f=$(ls -t "$out"/*/*kernel_stats.csv 2>/dev/null | head -1)
cut -c1-200 "$out/bench.json" | tail -1
[ -n "$f" ] && { cp "$f" "$out/kernel_stats.csv"; cut -d, -f1-5 "$f" | head -8; }
