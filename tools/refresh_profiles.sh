#!/bin/bash
# Regenerates the measured artifacts under profiles/ on the GPU box (run through gpurun; outputs land in
# gpurun_out/refresh/, copy them into profiles/ afterwards).  Counter passes are separate runs with
# --kernel-trace only, as the pool requires.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/refresh
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
echo "[1/6] bench default"; timeout -k 10 400 python3 "$R/bench.py" > "$O/bench_default.json" 2> "$O/bench_default.err" || echo "bench failed"
tail -c 400 "$O/bench_default.json"; echo
echo "[2/6] bench --pipeline"; timeout -k 10 300 python3 "$R/bench.py" --pipeline --no-cpu > "$O/bench_pipeline.json" 2> "$O/bench_pipeline.err" || echo "pipeline failed"
echo "[3/6] kernel trace"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/ktrace" -- python3 "$R/bench.py" --no-cpu > "$O/ktrace.json" 2> "$O/ktrace.err"; echo "rocprofv3 exit $?"
cp "$(ls -t "$O"/ktrace/*/*kernel_stats.csv | head -1)" "$O/bench_kernel_stats.csv"
for C in FETCH_SIZE WRITE_SIZE; do
  echo "[pmc] $C"; timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$O/pmc_bench_$C" -- python3 "$R/bench.py" --steps 20 --warmup 4 --no-cpu --no-events > /dev/null 2> "$O/pmc_$C.err"; echo "rocprofv3 exit $?"
done
python3 "$R/tools/pmc_summary.py" "$O" pmc_bench "$O/pmc_bench_f8x2M.json" && head -c 600 "$O/pmc_bench_f8x2M.json"; echo
echo "[6/6] probes"; timeout -k 10 200 python3 "$R/tools/frame100_bench.py" > "$O/frame100.json" 2> /dev/null; timeout -k 10 200 python3 "$R/tools/k2_probe.py" 20 2> /dev/null | tee "$O/k2_probe_20.txt"
cut -d, -f1-4 "$O/bench_kernel_stats.csv" | grep lpf_ | cut -c1-110
