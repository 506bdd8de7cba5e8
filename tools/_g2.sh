cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --steps 10 --warmup 3 --dist-backend gloo --force-device 0 --no-cpu > gpurun_out/r4_2rank.json 2> gpurun_out/r4_2rank.err; echo "exit $?"; tail -2 gpurun_out/r4_2rank.err | cut -c1-300
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r4_2rank.json').read().strip().splitlines()[-1])
print(d['n_gpus'], round(d['value']/1e9,1), round(1e3*d['ms_per_step'],2), d['config']['ranks'], d['config']['library'], d['scaling'])
PY
timeout -k 10 300 python -m pytest tests/test_resize.py tests/test_gpu_frame_step.py -x -q -m gpu 2>&1 | tail -2
