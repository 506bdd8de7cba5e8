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
