cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r4_t5.log 2>&1; echo "pytest exit $?"; tail -12 gpurun_out/r4_t5.log
timeout -k 10 200 python3 tools/stream_probe.py 1 2>&1 | grep "us per step"
timeout -k 10 200 python3 tools/frame100_bench.py 2>/dev/null | tail -1 | cut -c1-900
PROBE_RECTS=1 timeout -k 10 120 python3 tools/real_probe.py fused-pack rects 2>&1 | grep "us per step"
timeout -k 10 120 python3 tools/cloud_probe.py 2000000 rects both 2>&1 | grep "per step"
