cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tr_s20 -- python3 $GRAFT_REPO_ROOT/tools/real_probe.py serial rects 20 2>&1 | grep "us per step"
cut -d, -f1-4 $(ls -t $GRAFT_REPO_ROOT/gpurun_out/tr_s20/*/*kernel_stats.csv | head -1) | head -8 | cut -c1-140
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import os, sys, numpy as np, torch, time
sys.path.insert(0, '.')
exec(open('tools/real_probe.py').read().split("with LpfContext")[0].replace('mode = sys.argv[1] if len(sys.argv) > 1 else "serial"','mode="serial"').replace('lab = sys.argv[2] if len(sys.argv) > 2 else ""','lab="rects"').replace('nfr = int(sys.argv[3]) if len(sys.argv) > 3 else 146','nfr=20').replace('geometry = sys.argv[4] if len(sys.argv) > 4 else ""','geometry=""'))
with LpfContext(0) as ctx:
    ctx.set_camera(T, K, W, H, 0.0, 50.0)
    d_rects = torch.from_numpy(LpfContext.mask_rects(np.stack([f["masks"] for f in batch]))).to(dev)
    fn = ctx.make_device_step(d_pts, off, masks_u8=d_masks, lend=True, boxes_cam0=d_cam0, mask_rects=d_rects, box_off=boff, T_cam_to_velo=Tcv, inst_cap=cap, **o)
    for _ in range(5): fn()
    ctx.sync(); ctx.stats(reset=True)
    for _ in range(10): fn()
    print(ctx.stats()); ctx.sync()
PY
