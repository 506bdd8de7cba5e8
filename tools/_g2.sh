cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r4_t9.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/r4_t9.log
timeout -k 10 300 python3 tools/process_frames_bench.py 2>/dev/null | tail -1
