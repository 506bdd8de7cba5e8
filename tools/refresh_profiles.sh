#!/bin/bash
# Regenerates the measured artifacts under profiles/ on the GPU box (run through gpurun; outputs land in
# gpurun_out/refresh/, copy them into profiles/ afterwards).  Counter passes are separate runs with
# --kernel-trace only, as the pool requires.   usage: tools/refresh_profiles.sh [round tag, default r02] [pmc]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r02}
O=$R/gpurun_out/refresh
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
PMC_ONLY=${2:-}          # "pmc": only the counter passes and the bench lines that quote them (after a source change that moves no time)
if [ "$PMC_ONLY" = pmc ]; then
  for M in fused-pack serial; do for C in FETCH_SIZE WRITE_SIZE; do
    echo "[pmc] $M $C"; timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$O/pmc_${M}_$C" -- python3 "$R/bench.py" --mode $M --steps 20 --warmup 4 --no-cpu --no-secondary --no-events > /dev/null 2> "$O/pmc_${M}_$C.err"; echo "rocprofv3 exit $?"
  done; done
  python3 "$R/tools/pmc_summary.py" "$O" pmc_fused-pack pmc_serial "$O/${TAG}_pmc_bench_f8x2M.json" || exit 1
  cp "$O/${TAG}_pmc_bench_f8x2M.json" "$R/profiles/${TAG}_pmc_bench_f8x2M.json"
  timeout -k 10 400 python3 "$R/bench.py" > "$O/${TAG}_bench_default.json" 2> "$O/bench_default.err" || echo "bench failed"
  timeout -k 10 300 python3 "$R/bench.py" --gpus 1 --steps 20 --warmup 5 --no-secondary > "$O/${TAG}_bench_driver_args.json" 2> "$O/bench_driver_args.err" || echo "driver-args bench failed"
  tail -c 300 "$O/${TAG}_bench_driver_args.json"; echo
  exit 0
fi
echo "[1] bench default (fused-pack queueing) with the secondary configs and the CPU baseline"
timeout -k 10 400 python3 "$R/bench.py" > "$O/${TAG}_bench_default.json" 2> "$O/bench_default.err" || echo "bench failed"
tail -c 300 "$O/${TAG}_bench_default.json"; echo
echo "[2] bench --mode serial / pipeline"
timeout -k 10 300 python3 "$R/bench.py" --mode serial --no-cpu --no-secondary > "$O/${TAG}_bench_serial.json" 2> "$O/bench_serial.err" || echo "serial failed"
timeout -k 10 300 python3 "$R/bench.py" --mode pipeline --no-cpu --no-secondary > "$O/${TAG}_bench_pipeline.json" 2> "$O/bench_pipeline.err" || echo "pipeline failed"
timeout -k 10 300 python3 "$R/bench.py" --mode fused --no-cpu --no-secondary > "$O/${TAG}_bench_fused.json" 2> "$O/bench_fused.err" || echo "fused failed"
for M in fused-pack serial; do
  echo "[3] kernel trace, $M"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/ktrace_$M" -- python3 "$R/bench.py" --mode $M --no-cpu --no-secondary --no-events > "$O/ktrace_$M.json" 2> "$O/ktrace_$M.err"; echo "rocprofv3 exit $?"
  cp "$(ls -t "$O"/ktrace_$M/*/*kernel_stats.csv | head -1)" "$O/${TAG}_bench_kernel_stats_$M.csv"
  python3 "$R/tools/trace_gaps.py" "$(ls -t "$O"/ktrace_$M/*/*kernel_trace.csv | head -1)" $([ $M = fused-pack ] && echo lpf_step_t || echo lpf_k1_project) > "$O/${TAG}_timeline_$M.txt"
  head -14 "$O/${TAG}_timeline_$M.txt"
  for C in FETCH_SIZE WRITE_SIZE; do
    echo "[pmc] $M $C"; timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$O/pmc_${M}_$C" -- python3 "$R/bench.py" --mode $M --steps 20 --warmup 4 --no-cpu --no-secondary --no-events > /dev/null 2> "$O/pmc_${M}_$C.err"; echo "rocprofv3 exit $?"
  done
done
python3 "$R/tools/pmc_summary.py" "$O" pmc_fused-pack pmc_serial "$O/${TAG}_pmc_bench_f8x2M.json" && head -c 900 "$O/${TAG}_pmc_bench_f8x2M.json"; echo
echo "[3b] bench default again, now that the counter file of these sources exists (roofline.traffic)"
cp "$O/${TAG}_pmc_bench_f8x2M.json" "$R/profiles/${TAG}_pmc_bench_f8x2M.json"
timeout -k 10 400 python3 "$R/bench.py" > "$O/${TAG}_bench_default.json" 2> "$O/bench_default.err" || echo "bench failed"
timeout -k 10 300 python3 "$R/bench.py" --gpus 1 --steps 20 --warmup 5 --no-secondary > "$O/${TAG}_bench_driver_args.json" 2> "$O/bench_driver_args.err" || echo "driver-args bench failed"
echo "[4] probes"
timeout -k 10 200 python3 "$R/tools/frame100_bench.py" > "$O/${TAG}_frame100_configs01.json" 2> /dev/null
timeout -k 10 200 python3 "$R/tools/k2_probe.py" 20 2> /dev/null | tee "$O/${TAG}_k2_probe_20frames.txt"
timeout -k 10 200 python3 "$R/tools/stream_latency.py" --frames 120 > "$O/${TAG}_stream_latency_configs4.json" 2> /dev/null; cut -c1-300 "$O/${TAG}_stream_latency_configs4.json"
timeout -k 10 200 "$R/tools/mode_sweep.sh" full 2>/dev/null | tee "$O/${TAG}_mode_sweep.txt"
cut -d, -f1-4 "$O/${TAG}_bench_kernel_stats_fused-pack.csv" | grep lpf_ | cut -c1-110
cut -d, -f1-4 "$O/${TAG}_bench_kernel_stats_serial.csv" | grep lpf_ | cut -c1-110
