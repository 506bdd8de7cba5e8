#!/usr/bin/env python3
"""BASELINE.json configs[2] literally -- ONE synthetic cloud (default 2 000 000 points, 8 disk masks, 32 boxes, depth < 30) per launch
set, six resident clouds in turn, every step with its own masks and boxes: microseconds per step in order and as a software-
pipelined stream.  usage: python tools/cloud_probe.py [points] [rects|-] [serial|fused|fused-pack|both]
(rects: the masks' 2D rectangles are given -- lpf_set_mask_rects -- so the tiles read the masks themselves: no pack)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from lidar_object_detection_amd import synthetic as S  # noqa: E402
from lidar_object_detection_amd._native import LpfContext, SUMMARY_DTYPE  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
use_rects = len(sys.argv) > 2 and sys.argv[2] == "rects"
which = sys.argv[3] if len(sys.argv) > 3 else "both"
geometry = sys.argv[4] if len(sys.argv) > 4 else ""          # lab build: forced launch geometry (LpfContext.set_geometry)
M, B = 8, 32
dev = torch.device("cuda", 0)
TrVeloToCam, T, K, W, H = S.default_calibration()
Tcv = np.linalg.inv(TrVeloToCam)
bufs = []
for i in range(6):
    sc = S.scene(n, M, B, seed=7000 + i)
    o = dict(uv=torch.empty((n, 2), dtype=torch.int32, device=dev), label_bits=torch.empty(n, dtype=torch.int32, device=dev),
             valid_idx=torch.empty(n, dtype=torch.int64, device=dev), inst_idx=torch.empty((1, n), dtype=torch.int64, device=dev),
             count_mb=torch.zeros(M * B, dtype=torch.int32, device=dev), summary=torch.zeros(SUMMARY_DTYPE.itemsize, dtype=torch.uint8, device=dev))
    bufs.append((torch.from_numpy(sc["points"]).to(dev), torch.from_numpy(sc["masks"][None]).to(dev), o,
                 torch.from_numpy(np.ascontiguousarray(sc["corners_cam0"])).to(dev),
                 torch.from_numpy(LpfContext.mask_rects(sc["masks"][None])).to(dev) if use_rects else None))
torch.cuda.synchronize(dev)
res = []
modes = {"both": (False, "fused-pack"), "serial": (False,), "fused": ("fused",), "fused-pack": ("fused-pack",)}[which]
for mode in modes:
    from lidar_object_detection_amd import _build
    with LpfContext(0, library=_build.LAB_LIB if geometry else None) as ctx:
        if geometry:
            ctx.set_geometry(geometry)
        ctx.set_pipelined(mode)
        ctx.set_camera(T, K, W, H, 0.0, 30.0)
        fns = [ctx.make_device_step(p_, np.array([0, n], np.int64), masks_u8=m_, lend=True, boxes_cam0=c_, box_off=np.array([0, B], np.int32),
                                    T_cam_to_velo=Tcv, inst_cap=n, mask_rects=r_, **o) for p_, m_, o, c_, r_ in bufs]
        for _ in range(10):
            for f in fns:
                f()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(100):
            for f in fns:
                f()
        ctx.sync()
        res.append(1e6 * (time.perf_counter() - t0) / 600)
print("one %d-point cloud per launch set (8 masks, 32 boxes, own masks and boxes every step%s): %s" % (
    n, (", mask rectangles given" if use_rects else "") + (", geometry " + geometry if geometry else ""), ", ".join("%s %.1f us per step" % (m or "in order", r) for m, r in zip(modes, res))), flush=True)

# ---- the verdict's experiment, emulated with what exists: ONE cloud cut into k sub-ranges that are queued as k consecutive runs of a
#      software-pipelined context and drained at the end of the cloud (lpf_sync) -- the lists / box counts of range j ride among the
#      tiles of range j + 1, the last range's tail and the summaries come in the drain's launches.  Each range gets its own lists (a
#      real implementation would have to concatenate the instance lists afterwards: list m starts behind ALL of list m - 1), so this
#      is a lower bound of what sub-range pipelining inside one lpf_run could cost.
if which == "both" and not geometry:
    for k, pmode in ((2, "fused"), (4, "fused"), (2, "fused-pack")):
        with LpfContext(0) as ctx:
            ctx.set_pipelined(pmode)
            ctx.set_camera(T, K, W, H, 0.0, 30.0)
            fns = []
            for p_, m_, o, c_, r_ in bufs:
                part = n // k
                views = {kk: (v[:part] if v.shape[0] == n else v) for kk, v in o.items() if v is not None}      # (outputs of a range: the first part's buffers, reused)
                fns.append([ctx.make_device_step(p_[j * part:(j + 1) * part], np.array([0, part], np.int64), masks_u8=m_, lend=True, boxes_cam0=c_,
                                                 box_off=np.array([0, B], np.int32), T_cam_to_velo=Tcv, inst_cap=part,
                                                 **{kk: (v if kk != "inst_idx" else v[:, :part]) for kk, v in views.items()}) for j in range(k)])
            for _ in range(5):
                for parts in fns:
                    for f in parts:
                        f()
                    ctx.sync()
            t0 = time.perf_counter()
            for _ in range(50):
                for parts in fns:
                    for f in parts:
                        f()
                    ctx.sync()                              # the cloud's results complete: an in-order run returns them at this point
            print("the same cloud as %d sub-ranges, pipelined (%s) and drained per cloud: %.1f us per cloud (host sync included)" % (k, pmode, 1e6 * (time.perf_counter() - t0) / 300), flush=True)
    with LpfContext(0) as ctx:                              # reference point with the same host-side sync: in order, one run per cloud
        ctx.set_camera(T, K, W, H, 0.0, 30.0)
        fns = [ctx.make_device_step(p_, np.array([0, n], np.int64), masks_u8=m_, lend=True, boxes_cam0=c_, box_off=np.array([0, B], np.int32),
                                    T_cam_to_velo=Tcv, inst_cap=n, **o) for p_, m_, o, c_, r_ in bufs]
        for _ in range(5):
            for f in fns:
                f(); ctx.sync()
        t0 = time.perf_counter()
        for _ in range(50):
            for f in fns:
                f(); ctx.sync()
        print("in order, one run per cloud, synchronised per cloud: %.1f us per cloud (host sync included)" % (1e6 * (time.perf_counter() - t0) / 300), flush=True)

