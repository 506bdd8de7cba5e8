#!/usr/bin/env python3
"""Duration of the list/box-count kernel on real-frame-shaped input, by HIP events around whole steps with the
kernel form forced: frame 100 of the sample (109 355 points, 5 masks, 25 boxes), F copies batched.
usage: python tools/k2_probe.py [F] [full|nolists|noboxes|bare|validonly|instonly]   (drop the lists, the boxes or both: what each phase costs)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    from conftest import load_calib, load_golden, unpack_masks
    from lidar_object_detection_amd import _build
    from lidar_object_detection_amd._native import LpfContext, SUMMARY_DTYPE
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    variant = sys.argv[2] if len(sys.argv) > 2 else "full"
    calib = load_calib(); g = load_golden(100)
    W, H = int(calib["width"]), int(calib["height"])
    T, K3 = np.asarray(calib["TrVeloToRect"]), np.asarray(calib["K"])[:3, :3]
    pts = np.ascontiguousarray(g["points"]); masks = unpack_masks(g, "rect5", H, W).astype(np.uint8); corners = g["corners_velo"]
    n, M, B = len(pts), len(masks), len(corners)
    dev = torch.device("cuda", 0); stream = torch.cuda.Stream(dev)
    res = {}
    with torch.cuda.stream(stream):
        d_pts = torch.from_numpy(np.tile(pts, (F, 1))).to(dev)
        d_masks = torch.from_numpy(np.tile(masks[None], (F, 1, 1, 1))).to(dev)
        off = np.arange(F + 1, dtype=np.int64) * n
        o = dict(uv=torch.empty((F * n, 2), dtype=torch.int32, device=dev), label_bits=torch.empty(F * n, dtype=torch.int32, device=dev),
                 valid_idx=torch.empty(F * n, dtype=torch.int64, device=dev), inst_idx=torch.empty((F, n), dtype=torch.int64, device=dev),
                 count_mb=torch.zeros(F * M * B, dtype=torch.int32, device=dev),
                 summary=torch.zeros(F * SUMMARY_DTYPE.itemsize, dtype=torch.uint8, device=dev))
        if variant in ("nolists", "bare"):
            o.pop("valid_idx"); o.pop("inst_idx")
        if variant == "validonly":
            o.pop("inst_idx")
        if variant == "instonly":
            o.pop("valid_idx")
        for form in (os.environ.get("K2_FORMS", "small,large").split(",")):
            ctx = LpfContext(0, library=_build.LAB_LIB)      # forced geometries: the lab build
            ctx.set_stream(stream.cuda_stream)
            ctx.set_camera(T, K3, W, H, 0.0, 50.0)
            if variant not in ("noboxes", "bare", "validonly", "instonly"):
                ctx.set_boxes([corners] * F)
            ctx.set_geometry(form)
            if os.environ.get("K2_PIPE"):                    # e.g. fused-pack: a software-pipelined stream of such steps
                ctx.set_pipelined(os.environ["K2_PIPE"])
            step = ctx.make_device_step(d_pts, off, masks_u8=d_masks, erode_iters=0, lend=True, inst_cap=n, **o)
            for _ in range(20):
                step()
            ctx.sync()
            stream.synchronize()
            cm = o["count_mb"].cpu().numpy().reshape(F, M, B)
            assert variant in ("noboxes", "bare", "validonly", "instonly") or all(np.array_equal(cm[f], g["count_mb_rect5_d50"]) for f in range(F))
            t0 = time.perf_counter()
            for _ in range(500):
                step()
            ctx.sync()
            stream.synchronize()
            res[form] = (time.perf_counter() - t0) / 500 * 1e6
            ctx.close()
    print("%-8s F=%d frames of %d points: %s" % (variant, F, n, "  ".join("%s %.1f us/step" % kv for kv in res.items())))


if __name__ == "__main__":
    main()
