#!/bin/bash
# Counter passes on the real-scan step (tools/real_probe.py: 146 real frames, 16.9 M points per step) and, for comparison, on the
# synthetic headline step in order (bench.py --mode serial): HBM bytes (FETCH_SIZE / WRITE_SIZE), L2 hits and misses, L1 -> L2
# requests, wave cycles -- one counter group per pass, each pass a run of its own with --kernel-trace only (what the pool allows).
# Outputs land in gpurun_out/pmc_real/; tools/pmc_real_summary.py turns them into profiles/<tag>_pmc_real146.json.
# usage: tools/pmc_real.sh [tag, default r04] [probe variant, default rects] [modes, default "serial fused-pack"]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r04}
VAR=${2:-rects}
MODES=${3:-serial fused-pack}
O=$R/gpurun_out/pmc_real
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$O/counters_list.txt" 2>&1 || true
GROUPS_=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_ATOMIC_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU")
for M in $MODES; do
  echo "[trace] real_probe $M $VAR"
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace_real_${M}" -- python3 "$R/tools/real_probe.py" $M $VAR > "$O/real_${M}.txt" 2> "$O/real_${M}.err"; echo "exit $?"; cat "$O/real_${M}.txt"
  cp "$(ls -t "$O"/trace_real_${M}/*/*kernel_stats.csv | head -1)" "$O/${TAG}_real146_kernel_stats_${M}_${VAR}.csv"
  i=0
  for G in "${GROUPS_[@]}"; do
    echo "[pmc] real_probe $M $VAR: $G"
    timeout -k 10 200 rocprofv3 --pmc $G --kernel-trace --output-format csv -d "$O/pmc_real_${M}_g$i" -- python3 "$R/tools/real_probe.py" $M $VAR > /dev/null 2> "$O/pmc_real_${M}_g$i.err"; echo "exit $?"
    i=$((i + 1))
  done
done
if [ "${PMC_SYNTH:-1}" = 1 ]; then
  i=0
  for G in "${GROUPS_[@]}"; do
    echo "[pmc] bench serial: $G"
    timeout -k 10 300 rocprofv3 --pmc $G --kernel-trace --output-format csv -d "$O/pmc_synth_serial_g$i" -- python3 "$R/bench.py" --mode serial --steps 20 --warmup 4 --no-cpu --no-secondary --no-events > /dev/null 2> "$O/pmc_synth_serial_g$i.err"; echo "exit $?"
    i=$((i + 1))
  done
fi
python3 "$R/tools/pmc_real_summary.py" "$O" "$O/${TAG}_pmc_real146.json" || echo "summary failed"
