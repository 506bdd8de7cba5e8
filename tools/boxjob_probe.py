#!/usr/bin/env python3
"""Duration of the box job (lpf_box_job_kernel: per-frame box preparation + table set-up) as a kernel of its own, for a few
(frames, boxes per frame) shapes: set_boxes_cam0 leaves the job to the next run; lpf_release_to_stream launches it alone.
usage: python tools/boxjob_probe.py   (wrap in rocprofv3 --kernel-trace --stats for the kernel's own average)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from lidar_object_detection_amd import synthetic as S  # noqa: E402
from lidar_object_detection_amd._native import LpfContext  # noqa: E402

dev = torch.device("cuda", 0)
TrVeloToCam, T, K, W, H = S.default_calibration()
Tcv = np.linalg.inv(TrVeloToCam)
stream = torch.cuda.Stream(dev)
with torch.cuda.stream(stream), LpfContext(0) as ctx:
    ctx.set_stream(stream.cuda_stream)
    ctx.set_camera(T, K, W, H, 0.0, 50.0)

    def job():                                  # a box job waits for its run; lpf_release_to_stream launches what is owed, without waiting
        ctx.set_boxes_cam0_device(d, off, Tcv, lend=True)
        ctx.release_to_stream(stream.cuda_stream)

    for F, B in ((1, 32), (8, 32), (1, 314), (20, 31), (146, 136)):
        cam = np.concatenate([S.synthetic_boxes(B, seed=10 * f + B)[0] for f in range(F)])
        d = torch.from_numpy(np.ascontiguousarray(cam)).to(dev)
        off = np.arange(F + 1, dtype=np.int32) * B
        for _ in range(20):
            job()
        ctx.sync()
        t0 = time.perf_counter()
        reps = 500
        for _ in range(reps):
            job()
        t1 = time.perf_counter()
        ctx.sync()
        t2 = time.perf_counter()
        print("F=%3d B=%3d: %.2f us per call (host %.2f us per call)" % (F, B, 1e6 * (t2 - t0) / reps, 1e6 * (t1 - t0) / reps), flush=True)
