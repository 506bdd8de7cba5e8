"""LAB: timeline of one step on sample frame 100 from the per-block stamps of a build patched by tools/lab_stamps_apply.py."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import load_calib, load_golden, unpack_masks
from lidar_object_detection_amd._native import LpfContext, SUMMARY_DTYPE
out = os.path.join(ROOT, "gpurun_out", "stamps.bin")
os.makedirs(os.path.dirname(out), exist_ok=True)
os.environ["LAB_STAMPS"] = out
calib = load_calib(); g = load_golden(100)
W, H = int(calib["width"]), int(calib["height"])
T, K3 = np.asarray(calib["TrVeloToRect"]), np.asarray(calib["K"])[:3, :3]
pts = np.ascontiguousarray(g["points"]); masks = unpack_masks(g, "rect5", H, W).astype(np.uint8); corners = g["corners_velo"]
n, M, B = len(pts), len(masks), len(corners)
dev = torch.device("cuda", 0); stream = torch.cuda.Stream(dev)
with torch.cuda.stream(stream):
    d_pts = torch.from_numpy(pts).to(dev); d_masks = torch.from_numpy(masks[None]).to(dev)
    off = np.array([0, n], np.int64)
    o = dict(uv=torch.empty((n, 2), dtype=torch.int32, device=dev), label_bits=torch.empty(n, dtype=torch.int32, device=dev),
             valid_idx=torch.empty(n, dtype=torch.int64, device=dev), inst_idx=torch.empty((1, n), dtype=torch.int64, device=dev),
             count_mb=torch.zeros(M * B, dtype=torch.int32, device=dev), summary=torch.zeros(SUMMARY_DTYPE.itemsize, dtype=torch.uint8, device=dev))
    ctx = LpfContext(0); ctx.set_stream(stream.cuda_stream); ctx.set_camera(T, K3, W, H, 0.0, 50.0); ctx.set_boxes([corners]); ctx.set_geometry("small")
    step = ctx.make_device_step(d_pts, off, masks_u8=d_masks, erode_iters=0, inst_cap=n, **o)
    for _ in range(50): step()
    stream.synchronize()
    ctx.close()
d = np.fromfile(out, dtype=np.int64).reshape(-1, 8)
k1 = d[:1024]; tl = d[1024:2048]; fin = d[2048:2049]
k1 = k1[k1[:, 0] > 0]; tl = tl[tl[:, 0] > 0]
t0 = k1[:, 0].min()
us = lambda x: (x - t0) / 100.0
print("K1 blocks %d: start %.2f..%.2f  end %.2f..%.2f" % (len(k1), us(k1[:, 0].min()), us(k1[:, 0].max()), us(k1[:, 1].min()), us(k1[:, 1].max())))
cnt = tl[tl[:, 1] > 0]; lst = tl[tl[:, 4] > 0]
print("count blocks %d: start %.2f..%.2f staged %.2f..%.2f chunks done %.2f..%.2f flushed %.2f..%.2f" % (len(cnt), us(cnt[:, 0].min()), us(cnt[:, 0].max()), us(cnt[:, 1].min()), us(cnt[:, 1].max()), us(cnt[:, 2].min()), us(cnt[:, 2].max()), us(cnt[:, 3].min()), us(cnt[:, 3].max())))
for r in cnt[np.argsort(-cnt[:, 5])][:6]:
    print("   L=%5d start %.2f staged %.2f chunks %.2f flushed %.2f" % (r[5], us(r[0]), us(r[1]), us(r[2]), us(r[3])))
print("list blocks %d: start %.2f..%.2f end %.2f..%.2f (longest %.2f)" % (len(lst), us(lst[:, 0].min()), us(lst[:, 0].max()), us(lst[:, 4].min()), us(lst[:, 4].max()), ((lst[:, 4] - lst[:, 0]) / 100.0).max()))
print("finalize: start %.2f end %.2f" % (us(fin[0, 0]), us(fin[0, 1])))
