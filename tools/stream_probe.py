#!/usr/bin/env python3
"""A software-pipelined stream of single real frames (golden frame 100): microseconds per frame on the GPU and on the host (the
time the calls themselves take in a loop of thousands -- which includes waiting for queue space once the host runs ahead of the GPU, so
it approaches the GPU's figure from below when the stream is GPU-bound; tools/host_cost_probe.py times the calls' own CPU work), with boxes per
frame and with static boxes.
usage: python tools/stream_probe.py [frames-per-batch]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from lidar_object_detection_amd._native import LpfContext, SUMMARY_DTYPE  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda", 0)
g = np.load(os.path.join(ROOT, "tests", "golden", "frame_0000000100.npz"))
cal = np.load(os.path.join(ROOT, "tests", "golden", "calib_cam0.npz"))
T, K, W, H = np.asarray(cal["TrVeloToRect"]), np.asarray(cal["K"])[:3, :3], int(cal["width"]), int(cal["height"])
Tcv = np.linalg.inv(np.asarray(cal["TrVeloToCam"]))
pts = np.ascontiguousarray(g["points"], dtype=np.float32)
masks = np.unpackbits(g["masks_rect5_packed"], axis=-1)[..., :W].astype(np.uint8)
cam0 = np.ascontiguousarray(g["corners_cam0_raw"])
n, M, B = len(pts), len(masks), len(cam0)
d_pts = torch.from_numpy(np.tile(pts, (F, 1))).to(dev)
d_masks = torch.from_numpy(np.tile(masks[None], (F, 1, 1, 1))).to(dev)
d_cam0 = torch.from_numpy(np.tile(cam0, (F, 1, 1))).to(dev)
off = np.arange(F + 1, dtype=np.int64) * n
boff = np.arange(F + 1, dtype=np.int32) * B
o = dict(uv=torch.empty((F * n, 2), dtype=torch.int32, device=dev), label_bits=torch.empty(F * n, dtype=torch.int32, device=dev),
         valid_idx=torch.empty(F * n, dtype=torch.int64, device=dev), inst_idx=torch.empty((F, n), dtype=torch.int64, device=dev),
         count_mb=torch.zeros(F * M * B, dtype=torch.int32, device=dev), summary=torch.zeros(F * SUMMARY_DTYPE.itemsize, dtype=torch.uint8, device=dev))
torch.cuda.synchronize(dev)
for mode, per_frame_boxes, one_call in [(m_, b_, False) for m_ in ("fused-pack", "fused", False) for b_ in (True, False)] + [("fused-pack", True, True), ("fused", True, True)]:
    if True:
        with LpfContext(0) as ctx:
            ctx.set_pipelined(mode)
            ctx.set_camera(T, K, W, H, 0.0, 50.0)
            if not per_frame_boxes:
                ctx.set_boxes_cam0_device(d_cam0, boff, Tcv, lend=True)
            fn = ctx.make_device_step(d_pts, off, masks_u8=d_masks, lend=True, boxes_cam0=d_cam0 if per_frame_boxes else None, box_off=boff,
                                      T_cam_to_velo=Tcv, inst_cap=n, **o)
            if F == 1 and one_call:                         # lpf_run_frame: masks, boxes and the run in ONE C call
                fn = ctx.make_frame_step(d_pts, masks_u8=d_masks[0], boxes_cam0=d_cam0 if per_frame_boxes else None, T_cam_to_velo=Tcv, inst_cap=n, **o)
            for _ in range(50):
                fn()
            ctx.sync()
            reps = 3000
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            t1 = time.perf_counter()
            ctx.sync()
            t2 = time.perf_counter()
            print("mode %-10s boxes %-9s F=%d%s: %.2f us per step, host calls %.2f us per step" % (
                mode, "per step" if per_frame_boxes else "static", F, " ONE C call per frame (lpf_run_frame)" if (one_call and F == 1) else "",
                1e6 * (t2 - t0) / reps, 1e6 * (t1 - t0) / reps), flush=True)
