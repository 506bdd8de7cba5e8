#!/bin/bash
# rocprofv3 --kernel-trace --stats of tools/k2_probe.py (F copies of sample frame 100 per step).  usage: tools/prof_probe.sh <F>
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
F=${1:-1}
out=$R/gpurun_out/prof_probe_$F
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 "$R/tools/k2_probe.py" "$F" > "$out/probe.txt" 2> "$out/probe.err"
echo "rocprofv3 exit $?"; cat "$out/probe.txt"
f=$(ls -t "$out"/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && { cp "$f" "$out/kernel_stats.csv"; grep lpf_ "$f" | cut -d, -f1-4 | cut -c1-120; }
