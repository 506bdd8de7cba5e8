#!/usr/bin/env python3
"""In-order and pipelined microseconds per frame for each of the four full-size golden frames alone (their own masks and boxes,
boxes prepared per frame on the device).  usage: python tools/frames_probe.py [frame number: only that frame, e.g. under rocprofv3 --kernel-trace --stats] [static|noboxes|nolists]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from lidar_object_detection_amd._native import LpfContext, SUMMARY_DTYPE  # noqa: E402

dev = torch.device("cuda", 0)
gdir = os.path.join(ROOT, "tests", "golden")
cal = np.load(os.path.join(gdir, "calib_cam0.npz"))
T, K, W, H = np.asarray(cal["TrVeloToRect"]), np.asarray(cal["K"])[:3, :3], int(cal["width"]), int(cal["height"])
Tcv = np.linalg.inv(np.asarray(cal["TrVeloToCam"]))
only = sys.argv[1] if len(sys.argv) > 1 else ""
variant = sys.argv[2] if len(sys.argv) > 2 else ""      # "static": boxes set once; "noboxes": no boxes at all; "nolists": no index lists
for name in ("frame_0000000100.npz", "frame_0000001461_full.npz", "frame_0000002098_full.npz", "frame_0000002449_full.npz"):
    if only and only not in name:
        continue
    g = np.load(os.path.join(gdir, name))
    pts = torch.from_numpy(np.ascontiguousarray(g["points"], dtype=np.float32)).to(dev)
    masks = torch.from_numpy(np.unpackbits(g["masks_rect5_packed"], axis=-1)[..., :W].astype(np.uint8)[None]).to(dev)
    cam0 = torch.from_numpy(np.ascontiguousarray(g["corners_cam0_raw"], dtype=np.float64)).to(dev)
    n, M, B = pts.shape[0], masks.shape[1], cam0.shape[0]
    o = dict(uv=torch.empty((n, 2), dtype=torch.int32, device=dev), label_bits=torch.empty(n, dtype=torch.int32, device=dev),
             valid_idx=torch.empty(n, dtype=torch.int64, device=dev), inst_idx=torch.empty((1, n), dtype=torch.int64, device=dev),
             count_mb=torch.zeros(M * B, dtype=torch.int32, device=dev), summary=torch.zeros(SUMMARY_DTYPE.itemsize, dtype=torch.uint8, device=dev))
    torch.cuda.synchronize(dev)
    res = []
    for mode in (False, "fused-pack"):
        with LpfContext(0) as ctx:
            ctx.set_pipelined(mode)
            ctx.set_camera(T, K, W, H, 0.0, 50.0)
            oo = dict(o)
            if variant == "nolists":
                oo["valid_idx"] = None; oo["inst_idx"] = None
            if variant == "static":
                ctx.set_boxes_cam0_device(cam0, np.array([0, B], np.int32), Tcv, lend=True)
            fn = ctx.make_device_step(pts, np.array([0, n], np.int64), masks_u8=masks, lend=True,
                                      boxes_cam0=cam0 if variant not in ("static", "noboxes") else None, box_off=np.array([0, B], np.int32),
                                      T_cam_to_velo=Tcv, inst_cap=n, **oo)
            for _ in range(30):
                fn()
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(1000):
                fn()
            ctx.sync()
            res.append(1e3 * (time.perf_counter() - t0))
    print("%s %s: N=%d boxes %d (kept %d): in order %.1f us, pipelined stream %.1f us per frame" % (name[:16], variant, n, B, len(g["visible_pos"]), res[0], res[1]), flush=True)
