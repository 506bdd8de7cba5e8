#!/bin/bash
# Regenerates the measured artifacts under profiles/ on the GPU box (run through gpurun; outputs land in
# gpurun_out/refresh/, copy them into profiles/ afterwards).  Counter passes are separate runs with
# --kernel-trace only, as the pool requires.   usage: tools/refresh_profiles.sh [round tag, default r04] [pmc|quick]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r04}
WHAT=${2:-all}           # "pmc": only the counter passes and the bench lines that quote them; "quick": no probes
O=$R/gpurun_out/refresh
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
pmc() {
  for M in fused-pack serial; do for C in FETCH_SIZE WRITE_SIZE; do
    echo "[pmc] $M $C"; timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$O/pmc_${M}_$C" -- python3 "$R/bench.py" --mode $M --steps 20 --warmup 4 --no-cpu --no-secondary --no-events > /dev/null 2> "$O/pmc_${M}_$C.err"; echo "rocprofv3 exit $?"
  done; done
  python3 "$R/tools/pmc_summary.py" "$O" pmc_fused-pack pmc_serial "$O/${TAG}_pmc_bench_f8x2M.json" || return 1
  cp "$O/${TAG}_pmc_bench_f8x2M.json" "$R/profiles/${TAG}_pmc_bench_f8x2M.json"
  head -c 900 "$O/${TAG}_pmc_bench_f8x2M.json"; echo
}
lines() {
  echo "[bench] default arguments, then the driver's"
  timeout -k 10 500 python3 "$R/bench.py" > "$O/${TAG}_bench_default.json" 2> "$O/bench_default.err" || echo "bench failed"
  timeout -k 10 300 python3 "$R/bench.py" --gpus 1 --steps 20 --warmup 5 > "$O/${TAG}_bench_driver_args.json" 2> "$O/bench_driver_args.err" || echo "driver-args bench failed"
  python3 -c "
import json
for n in ('default', 'driver_args'):
    d = json.load(open('$O/${TAG}_bench_%s.json' % n)); r = d['roofline']
    print('%-12s %.2f us/step  %.1f G points/s  bracket %.2f us  frac %.3f  traffic %s' % (n, 1e3 * d['ms_per_step'], d['value'] / 1e9, r['avg_us'], r['frac'], r['traffic']))"
}
if [ "$WHAT" = pmc ]; then pmc && lines; exit 0; fi
echo "[1] bench --mode serial / fused"
timeout -k 10 300 python3 "$R/bench.py" --mode serial --no-cpu --no-secondary > "$O/${TAG}_bench_serial.json" 2> "$O/bench_serial.err" || echo "serial failed"
timeout -k 10 300 python3 "$R/bench.py" --mode fused --no-cpu --no-secondary > "$O/${TAG}_bench_fused.json" 2> "$O/bench_fused.err" || echo "fused failed"
timeout -k 10 300 python3 "$R/bench.py" --static-boxes --no-cpu --no-secondary > "$O/${TAG}_bench_static_boxes.json" 2> "$O/bench_static.err" || echo "static-boxes failed"
for M in fused-pack serial; do
  echo "[2] kernel trace, $M"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/ktrace_$M" -- python3 "$R/bench.py" --mode $M --no-cpu --no-secondary --no-events > "$O/ktrace_$M.json" 2> "$O/ktrace_$M.err"; echo "rocprofv3 exit $?"
  cp "$(ls -t "$O"/ktrace_$M/*/*kernel_stats.csv | head -1)" "$O/${TAG}_bench_kernel_stats_$M.csv"
  python3 "$R/tools/trace_gaps.py" "$(ls -t "$O"/ktrace_$M/*/*kernel_trace.csv | head -1)" $([ $M = fused-pack ] && echo lpf_step_t || echo lpf_k1_project) > "$O/${TAG}_timeline_$M.txt"
  head -14 "$O/${TAG}_timeline_$M.txt"
done
echo "[2b] kernel trace at the driver's arguments (fill and drain of the pipeline)"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$O/ktrace_driver" -- python3 "$R/bench.py" --steps 20 --warmup 5 --no-cpu --no-secondary --no-events > /dev/null 2> "$O/ktrace_driver.err"
python3 "$R/tools/trace_list.py" "$(ls -t "$O"/ktrace_driver/*/*kernel_trace.csv | head -1)" 26 > "$O/${TAG}_timeline_driver_args.txt"; tail -6 "$O/${TAG}_timeline_driver_args.txt"
echo "[2c] real scans at the headline's size (146 frames per step) with the masks' rectangles: kernels in order, and the fused step; then without them"
for M in serial fused-pack; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/ktrace_real_$M" -- python3 "$R/tools/real_probe.py" $M rects > "$O/real_$M.txt" 2> /dev/null
  cp "$(ls -t "$O"/ktrace_real_$M/*/*kernel_stats.csv | head -1)" "$O/${TAG}_real146_kernel_stats_${M}_mask_rects.csv"; cat "$O/real_$M.txt"
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/ktrace_real_norects" -- python3 "$R/tools/real_probe.py" fused-pack > "$O/real_norects.txt" 2> /dev/null
cp "$(ls -t "$O"/ktrace_real_norects/*/*kernel_stats.csv | head -1)" "$O/${TAG}_real146_kernel_stats_fused-pack_no_rects.csv"; cat "$O/real_norects.txt"
echo "[2d] BASELINE configs[2] literally (one 2 M-point cloud per launch set): timelines in order and pipelined"
for M in serial fused-pack; do
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$O/ktrace_cloud_$M" -- python3 "$R/tools/cloud_probe.py" 2000000 - $M > "$O/cloud_$M.txt" 2> /dev/null
  { cat "$O/cloud_$M.txt"; python3 "$R/tools/trace_list.py" "$(ls -t "$O"/ktrace_cloud_$M/*/*kernel_trace.csv | head -1)" 12; } > "$O/${TAG}_timeline_configs2_one_cloud_$M.txt"; cat "$O/${TAG}_timeline_configs2_one_cloud_$M.txt"
done
echo "[2e] counter passes on the real-scan step and the synthetic step in order"
( cd "$R" && PMC_SYNTH=1 bash tools/pmc_real.sh $TAG rects > "$O/pmc_real.log" 2>&1; tail -14 "$O/pmc_real.log"; cp gpurun_out/pmc_real/${TAG}_pmc_real146.json "$O/" )
cd /tmp
pmc
lines
[ "$WHAT" = quick ] && exit 0
echo "[3] probes"
timeout -k 10 200 python3 "$R/tools/frame100_bench.py" > "$O/${TAG}_frame100_configs01.json" 2> /dev/null; cut -c1-400 "$O/${TAG}_frame100_configs01.json"
timeout -k 10 200 python3 "$R/tools/stream_latency.py" --frames 600 > "$O/${TAG}_stream_latency_configs4.json" 2> /dev/null; cut -c1-300 "$O/${TAG}_stream_latency_configs4.json"
{ timeout -k 10 200 python3 "$R/tools/boxjob_probe.py"; timeout -k 10 200 python3 "$R/tools/stream_probe.py" 1; timeout -k 10 200 python3 "$R/tools/stream_probe.py" 20; timeout -k 10 200 python3 "$R/tools/frames_probe.py"; timeout -k 10 200 python3 "$R/tools/cloud_probe.py" 2000000; } 2> /dev/null | tee "$O/${TAG}_probes.txt"
timeout -k 10 300 python3 "$R/tools/process_frames_bench.py" > "$O/${TAG}_process_frames.json" 2> /dev/null; cut -c1-400 "$O/${TAG}_process_frames.json"
cut -d, -f1-4 "$O/${TAG}_bench_kernel_stats_fused-pack.csv" | grep lpf_ | cut -c1-110
cut -d, -f1-4 "$O/${TAG}_bench_kernel_stats_serial.csv" | grep lpf_ | cut -c1-110
