#!/usr/bin/env python3
"""The headline step (8 x 2 M synthetic points, 8 masks, 32 boxes per cloud, masks lent every step, boxes set ONCE) through raw
ctypes, on any build of liblpf.so whose ABI has lpf_set_pipelined(4) (ABI 4 and later): same-box A/B of library builds.
usage: python tools/ab_step.py <liblpf.so> [<liblpf.so> ...]      (each library is timed three times, interleaved)"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from lidar_object_detection_amd import synthetic as S  # noqa: E402

P, I64 = ctypes.c_void_p, ctypes.c_int64


class Outputs(ctypes.Structure):
    _fields_ = [("uv", P), ("label_bits", P), ("depth", P), ("u_f", P), ("v_f", P), ("valid_idx", P), ("inst_idx", P), ("inst_cap", I64),
                ("count_mb", P), ("summary", P), ("on_device", ctypes.c_int32), ("reserved", ctypes.c_int32), ("uv_valid", P), ("label_valid", P)]


def main():
    dev = torch.device("cuda", 0)
    F, n, M, B = 8, 2_000_000, 8, 32
    _, T, K, W, H = S.default_calibration()
    scenes = [S.scene(n, M, B, seed=f) for f in range(F)]
    base = torch.from_numpy(np.concatenate([sc["points"] for sc in scenes])).to(dev)
    masks0 = torch.from_numpy(np.stack([sc["masks"] for sc in scenes])).to(dev)
    bufs = []
    for b in range(4):
        pts = base if b == 0 else base[torch.cat([f * n + torch.randperm(n, device=dev) for f in range(F)])].contiguous()
        o = dict(uv=torch.empty((F * n, 2), dtype=torch.int32, device=dev), lab=torch.empty(F * n, dtype=torch.int32, device=dev),
                 vidx=torch.empty(F * n, dtype=torch.int64, device=dev), iidx=torch.empty(F * n, dtype=torch.int64, device=dev),
                 cnt=torch.zeros(F * M * B, dtype=torch.int32, device=dev), summ=torch.zeros(F * 928, dtype=torch.uint8, device=dev))
        out = Outputs(uv=o["uv"].data_ptr(), label_bits=o["lab"].data_ptr(), valid_idx=o["vidx"].data_ptr(), inst_idx=o["iidx"].data_ptr(), inst_cap=n,
                      count_mb=o["cnt"].data_ptr(), summary=o["summ"].data_ptr(), on_device=1)
        bufs.append((pts, masks0.clone() if b else masks0, o, out))
    torch.cuda.synchronize(dev)
    off = (np.arange(F + 1, dtype=np.int64) * n)
    boff = (np.arange(F + 1, dtype=np.int32) * B)
    corners = np.ascontiguousarray(np.concatenate([sc["corners_velo"] for sc in scenes]))
    Tm = np.ascontiguousarray(T, np.float64)
    Km = np.ascontiguousarray(np.asarray(K)[:3, :3], np.float64)
    libs = []
    for path in sys.argv[1:]:
        lib = ctypes.CDLL(os.path.abspath(path))
        lib.lpf_create.argtypes = [ctypes.POINTER(P), ctypes.c_int]
        lib.lpf_set_camera.argtypes = [P, P, P, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double]
        lib.lpf_set_boxes.argtypes = [P, P, P, ctypes.c_int, ctypes.c_int]
        lib.lpf_set_masks_u8.argtypes = [P, P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        lib.lpf_run_batch.argtypes = [P, P, P, ctypes.c_int, ctypes.c_int, ctypes.POINTER(Outputs)]
        lib.lpf_set_pipelined.argtypes = [P, ctypes.c_int]
        lib.lpf_sync.argtypes = [P]
        lib.lpf_destroy.argtypes = [P]
        ctx = P()
        assert lib.lpf_create(ctypes.byref(ctx), 0) == 0
        assert lib.lpf_set_pipelined(ctx, 4) == 0
        assert lib.lpf_set_camera(ctx, Tm.ctypes.data, Km.ctypes.data, W, H, 0.0, 30.0) == 0
        assert lib.lpf_set_boxes(ctx, corners.ctypes.data, boff.ctypes.data, F, 1) == 0
        libs.append((path, lib, ctx))

    def steps(lib, ctx, k):
        for i in range(k):
            pts, m, o, out = bufs[i % 4]
            assert lib.lpf_set_masks_u8(ctx, m.data_ptr(), F, M, 0, 2) == 0
            assert lib.lpf_run_batch(ctx, pts.data_ptr(), off.ctypes.data, F, 1, ctypes.byref(out)) == 0
        assert lib.lpf_sync(ctx) == 0

    for path, lib, ctx in libs:
        steps(lib, ctx, 30)
    for rep in range(3):
        for path, lib, ctx in libs:
            for k in (300, 20):
                t0 = time.perf_counter()
                steps(lib, ctx, k)
                print("%-40s K=%3d  %.2f us per step" % (os.path.basename(path), k, 1e6 * (time.perf_counter() - t0) / k), flush=True)
    for path, lib, ctx in libs:
        lib.lpf_destroy(ctx)


if __name__ == "__main__":
    main()
