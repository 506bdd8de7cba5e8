#!/usr/bin/env python3
"""146 real frames per step (the four full-size golden frames in turn, ~16.9 M points): in-order steps for a kernel trace.
usage: rocprofv3 --kernel-trace --stats ... -- python3 tools/real_probe.py [serial|fused|fused-pack] [noboxes|nolists|nomasks|rects|-] [frames] [geometry: lab build]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from lidar_object_detection_amd._native import LpfContext, SUMMARY_DTYPE  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "serial"
lab = sys.argv[2] if len(sys.argv) > 2 else ""
nfr = int(sys.argv[3]) if len(sys.argv) > 3 else 146
geometry = sys.argv[4] if len(sys.argv) > 4 else ""
if lab == "-":
    lab = ""
dev = torch.device("cuda", 0)
gdir = os.path.join(ROOT, "tests", "golden")
cal = np.load(os.path.join(gdir, "calib_cam0.npz"))
T, K, W, H = np.asarray(cal["TrVeloToRect"]), np.asarray(cal["K"])[:3, :3], int(cal["width"]), int(cal["height"])
Tcv = np.linalg.inv(np.asarray(cal["TrVeloToCam"]))
frames = []
for name in ("frame_0000000100.npz", "frame_0000001461_full.npz", "frame_0000002098_full.npz", "frame_0000002449_full.npz"):
    g = np.load(os.path.join(gdir, name))
    frames.append(dict(points=np.ascontiguousarray(g["points"], dtype=np.float32),
                       masks=np.unpackbits(g["masks_rect5_packed"], axis=-1)[..., :W].astype(np.uint8),
                       cam0=np.ascontiguousarray(g["corners_cam0_raw"], dtype=np.float64)))
if os.environ.get("PROBE_ALIGN"):                         # experiment: every frame padded to a multiple of PROBE_ALIGN points (a far point behind the camera)
    q = int(os.environ["PROBE_ALIGN"])
    for f in frames:
        pad = (-len(f["points"])) % q
        f["points"] = np.concatenate([f["points"], np.tile(np.array([[-500.0, 0.0, 0.0, 0.0]], np.float32), (pad, 1))])
batch = [frames[i % 4] for i in range(nfr)]
sizes = [len(f["points"]) for f in batch]
off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
boff = np.concatenate([[0], np.cumsum([len(f["cam0"]) for f in batch])]).astype(np.int32)
M, ntot, cap = 5, int(off[-1]), max(sizes)
d_pts = torch.from_numpy(np.concatenate([f["points"] for f in batch])).to(dev)
d_masks = torch.from_numpy(np.stack([f["masks"] for f in batch])).to(dev)
d_cam0 = torch.from_numpy(np.concatenate([f["cam0"] for f in batch])).to(dev)
o = dict(uv=torch.empty((ntot, 2), dtype=torch.int32, device=dev), label_bits=torch.empty(ntot, dtype=torch.int32, device=dev),
         valid_idx=None if lab == "nolists" else torch.empty(ntot, dtype=torch.int64, device=dev),
         inst_idx=None if lab == "nolists" else torch.empty((nfr, cap), dtype=torch.int64, device=dev),
         count_mb=torch.zeros(M * int(boff[-1]), dtype=torch.int32, device=dev), summary=torch.zeros(nfr * SUMMARY_DTYPE.itemsize, dtype=torch.uint8, device=dev))
torch.cuda.synchronize(dev)
from lidar_object_detection_amd import _build  # noqa: E402
clock = bool(os.environ.get("ROLE_CLOCK"))          # lab build: per role of the step launches, block times (fused modes)
with LpfContext(0, library=os.environ.get("PROBE_LIBRARY") or (_build.LAB_LIB if (geometry or clock) else None)) as ctx:
    if geometry:
        ctx.set_geometry(geometry)
    ctx.set_pipelined(False if mode == "serial" else mode)
    ctx.set_camera(T, K, W, H, 0.0, 50.0)
    d_rects = torch.from_numpy(LpfContext.mask_rects(np.stack([f["masks"] for f in batch]))).to(dev) if (lab == "rects" or os.environ.get("PROBE_RECTS")) else None     # the masks' 2D rectangles (lpf_set_mask_rects)
    if d_rects is not None and os.environ.get("PROBE_EMPTY_RECTS"):      # experiment: every rectangle empty -- the tiles test them, nothing is ever inside
        d_rects.zero_()
    fn = ctx.make_device_step(d_pts, off, masks_u8=None if lab == "nomasks" else d_masks, lend=True, boxes_cam0=None if lab == "noboxes" else d_cam0, mask_rects=d_rects,
                              box_off=boff, T_cam_to_velo=Tcv, inst_cap=cap, **o)
    for _ in range(5):
        fn()
    ctx.sync()
    t0 = time.perf_counter()
    reps = 30
    for _ in range(reps):
        fn()
    ctx.sync()
    print("%s %s %s frames=%d points=%d: %.1f us per step" % (mode, lab or "all", geometry, nfr, ntot, 1e6 * (time.perf_counter() - t0) / reps), flush=True)
    if clock and mode != "serial":
        ctx.role_clock(reset=True)
        for _ in range(reps):
            fn()
        for role, r in ctx.role_clock(reset=True).items():
            print("  %-20s %7d blocks per launch, span %7.2f us, mean block %6.2f us, longest block %6.2f us" % (role, r["blocks"] // reps, r["span_us"] / 1.0, r["mean_us"], r["longest_us"]))
    sm = np.frombuffer(o["summary"].cpu().numpy().tobytes(), SUMMARY_DTYPE)
    print("valid %d masked %d list entries %d" % (sm["n_valid"].sum(), sm["n_labelled"].sum(), sm["inst_count"].sum()))
