#!/usr/bin/env python3
"""Where the host's time per frame goes in a software-pipelined stream of single real frames (golden frame 100): each of the three C
calls of a step timed on its own (ctypes, pre-marshalled arguments).  usage: python tools/host_cost_probe.py [frames per step]"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from lidar_object_detection_amd._native import LpfContext, Outputs, SUMMARY_DTYPE, _P  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda", 0)
g = np.load(os.path.join(ROOT, "tests", "golden", "frame_0000000100.npz"))
cal = np.load(os.path.join(ROOT, "tests", "golden", "calib_cam0.npz"))
T, K, W, H = np.asarray(cal["TrVeloToRect"]), np.asarray(cal["K"])[:3, :3], int(cal["width"]), int(cal["height"])
Tcv = np.ascontiguousarray(np.linalg.inv(np.asarray(cal["TrVeloToCam"]))).reshape(16)
n1 = len(g["points"])
pts = torch.from_numpy(np.tile(np.ascontiguousarray(g["points"], dtype=np.float32), (F, 1))).to(dev)
m1 = np.unpackbits(g["masks_rect5_packed"], axis=-1)[..., :W].astype(np.uint8)
masks = torch.from_numpy(np.tile(m1[None], (F, 1, 1, 1))).to(dev)
cam0 = torch.from_numpy(np.tile(np.ascontiguousarray(g["corners_cam0_raw"]), (F, 1, 1))).to(dev)
n, M, B = pts.shape[0], m1.shape[0], cam0.shape[0]
B1 = B // F
o = Outputs()
keep = dict(uv=torch.empty((n, 2), dtype=torch.int32, device=dev), lab=torch.empty(n, dtype=torch.int32, device=dev), vidx=torch.empty(n, dtype=torch.int64, device=dev),
            iidx=torch.empty((F, n1), dtype=torch.int64, device=dev), cnt=torch.zeros(M * B, dtype=torch.int32, device=dev), summ=torch.zeros(F * SUMMARY_DTYPE.itemsize, dtype=torch.uint8, device=dev))
o.uv, o.label_bits, o.valid_idx, o.inst_idx, o.inst_cap, o.count_mb, o.summary, o.on_device = (keep["uv"].data_ptr(), keep["lab"].data_ptr(), keep["vidx"].data_ptr(),
                                                                                          keep["iidx"].data_ptr(), n1, keep["cnt"].data_ptr(), keep["summ"].data_ptr(), 1)
boff = (np.arange(F + 1) * B1).astype(np.int32)
off = (np.arange(F + 1) * n1).astype(np.int64)
torch.cuda.synchronize(dev)
with LpfContext(0) as ctx:
    lib, h = ctx._lib, ctx._h
    ctx.set_pipelined("fused-pack")
    ctx.set_camera(T, K, W, H, 0.0, 50.0)
    a_m = (h, _P(masks.data_ptr()), F, M, 0, 2)
    a_b = (h, _P(cam0.data_ptr()), 2, _P(boff.ctypes.data), F, _P(Tcv.ctypes.data), 1, 1, None, None, None, None)
    a_r = (h, _P(pts.data_ptr()), _P(off.ctypes.data), F, 1, ctypes.byref(o))
    lib.lpf_run = lib.lpf_run_batch
    for _ in range(200):
        lib.lpf_set_masks_u8(*a_m); lib.lpf_set_boxes_cam0(*a_b); lib.lpf_run(*a_r)
    ctx.sync()
    # bursts of 32 steps from an idle stream, a sync in between: the queue never fills, so this is the calls' own CPU time (a loop
    # of thousands runs into the GPU's pace as soon as the host is the faster of the two)
    reps, burst = 3200, 32
    tm = tb = tr = 0.0
    pc = time.perf_counter
    for _ in range(reps // burst):
        for _ in range(burst):
            t0 = pc(); lib.lpf_set_masks_u8(*a_m); t1 = pc(); lib.lpf_set_boxes_cam0(*a_b); t2 = pc(); lib.lpf_run(*a_r); t3 = pc()
            tm += t1 - t0; tb += t2 - t1; tr += t3 - t2
        ctx.sync()
    t0 = pc()
    for _ in range(reps):
        pc(); pc(); pc(); pc()
    clk = (pc() - t0) / reps
    t0 = pc()
    for _ in range(reps):
        lib.lpf_abi_version()
    ffi = (pc() - t0) / reps
    print("F=%d frames per step -- " % F, end="")
    print("per step on the host: lpf_set_masks_u8 %.2f us, lpf_set_boxes_cam0 %.2f us, lpf_run_batch %.2f us (4 clock reads %.2f us; an empty C call %.2f us)" % (
        1e6 * tm / reps, 1e6 * tb / reps, 1e6 * tr / reps, 1e6 * clk, 1e6 * ffi))
