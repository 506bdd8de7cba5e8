"""Seeded differential fuzz of the whole HIP path against the CPU oracle: random frame counts and sizes, mask
counts and densities (including full masks -> dense segments), box counts on both sides of the 64-box candidate
word, oriented / axis-aligned tests, depth windows, and clouds concentrated inside the camera frustum (dense valid
runs, as real scans have) -- on the product library (geometry by launch size) and, on the lab build, under every forced
launch geometry (lpf_set_geometry: the code paths of large launches at small sizes)."""
import os

import numpy as np
import pytest

from lidar_object_detection_amd import synthetic as S
from oracle import cpu_oracle as orc

pytestmark = pytest.mark.gpu


def _frustum_cloud(rng, n, T, K, W, H, frac_in):
    """n points, about frac_in of them placed through the inverse projection inside the image (scan-order runs)."""
    pts = S.synthetic_cloud(max(n, 1), seed=int(rng.integers(1 << 30)))[:n]
    k = int(n * frac_in)
    if k:
        u = np.sort(rng.uniform(-40, W + 40, k))                         # sorted: neighbours in the array are neighbours in the image
        v = rng.uniform(-20, H + 20, k)
        d = rng.uniform(0.5, 70.0, k)
        cam = np.stack([(u - K[0, 2]) * d / K[0, 0], (v - K[1, 2]) * d / K[1, 1], d, np.ones(k)])
        velo = np.linalg.solve(T, cam)[:3].T
        start = int(rng.integers(0, n - k + 1))
        pts[start:start + k, :3] = velo.astype(np.float32)
    return pts


def _case(seed, calib):
    rng = np.random.default_rng(seed)
    _, T, K, W, H = S.default_calibration(calib)
    F = int(rng.integers(1, 5))
    M = int(rng.choice([0, 1, 3, 8, 9, 17, 32]))
    oriented = bool(rng.integers(0, 2))
    dmax = float(rng.choice([30.0, 50.0, 80.0]))
    frames, masks, boxes = [], [], []
    for f in range(F):
        n = int(rng.choice([0, 1, 63, 64, 65, 700, 4096, 4097, 9000, 20000]))
        frames.append(_frustum_cloud(rng, n, T, K, W, H, float(rng.choice([0.0, 0.3, 0.9]))))
        m = np.zeros((M, H, W), np.uint8)
        for i in range(M):
            kind = rng.integers(0, 4)
            if kind == 0:
                m[i] = 1                                                    # full mask: every valid point is masked
            elif kind == 1:
                m[i] = (rng.random((H, W)) < 0.5)
            elif kind == 2:
                x0, y0 = int(rng.integers(0, W - 50)), int(rng.integers(0, H - 50))
                m[i, y0:y0 + int(rng.integers(10, 200)), x0:x0 + int(rng.integers(10, 600))] = 1
        masks.append(m)
        B = int(rng.choice([0, 1, 7, 33, 64, 65, 130]))
        boxes.append(S.synthetic_boxes(B, seed=int(rng.integers(1 << 30)))[1] if B else np.zeros((0, 8, 3)))
    return T, K, W, H, dmax, oriented, M, frames, masks, boxes


@pytest.mark.parametrize("form", ["auto", "small-narrow", "small-1024", "large", "large-scan"])
@pytest.mark.parametrize("seed", range(int(os.environ.get("LPF_FUZZ_CASES", "24"))))
def test_fuzz_against_oracle(seed, form, calib):
    from conftest import context_for_form
    T, K, W, H, dmax, oriented, M, frames, masks, boxes = _case(int(os.environ.get("LPF_FUZZ_SEED_BASE", "1000")) + seed, calib)
    with context_for_form(form) as ctx:
        ctx.set_camera(T, K, W, H, 0.0, dmax)
        ctx.set_masks(np.stack(masks))
        ctx.set_boxes(boxes, oriented=oriented)
        res = ctx.run_batch(frames, want_float=True, want_valid_uv=True)
    for f, r in enumerate(res):
        lab = orc.pack_masks(masks[f], 0, H, W) if M else None
        o = orc.run(frames[f], T, K, W, H, 0.0, dmax, label_img=lab, M=M, corners=boxes[f], oriented=oriented)
        for k in ("u", "v", "label_bits", "valid_idx", "count_mb", "best_box", "best_cnt", "inst_count"):
            assert np.array_equal(r[k], o[k]), (seed, form, f, k)
        assert r["n_valid"] == o["n_valid"]
        assert np.array_equal(r["u_valid"], o["u"][o["valid_idx"]]) and np.array_equal(r["v_valid"], o["v"][o["valid_idx"]])
        assert np.array_equal(r["label_valid"], o["label_bits"][o["valid_idx"]])
        for a, b in zip(r["inst_lists"], o["inst_lists"]):
            assert np.array_equal(a, b), (seed, form, f)
        for k in ("depth", "uf", "vf"):
            assert np.array_equal(r[k], o[k], equal_nan=True), (seed, form, f, k)


@pytest.mark.parametrize("mode", ["fused", "fused+lent", "fused-pack+lent", "fused+lent+rects", "fused-pack+lent+rects", "fused-pack+lent+rects+large", "off+lent+rects+large"])
@pytest.mark.parametrize("seed", range(int(os.environ.get("LPF_FUZZ_CASES", "24")) // 2))
def test_fuzz_software_pipelined_device_mode(seed, mode, calib):
    """The same random cases through device mode under lpf_set_pipelined(2 / 4), three consecutive cases per context with nothing
    synchronised in between: the tail of case k rides in the launch of case k+1 (another shape, other masks and boxes), its
    summaries in the launch of case k+2.  "+rects": the masks' tight rectangles are given (lpf_set_mask_rects) -- results must not
    change; "+large" forces the large launch geometry (lab build), under which the tiles read the lent masks inside the rectangles
    through the candidate grid (LpfDirectRect; the grid rides in mode 4, goes ahead as a kernel in order)."""
    import torch
    from conftest import context_for_form
    from lidar_object_detection_amd._native import LpfContext, SUMMARY_DTYPE
    dev = torch.device("cuda", 0)
    base = int(os.environ.get("LPF_FUZZ_SEED_BASE", "1000"))
    cases = [_case(base + 3 * seed + j, calib) for j in range(3)]
    held = []
    with context_for_form("large" if "+large" in mode else "auto") as ctx:
        ctx.set_pipelined(False if mode.startswith("off") else mode.split("+")[0])
        for T, K, W, H, dmax, oriented, M, frames, masks, boxes in cases:
            F = len(frames)
            sizes = [len(p) for p in frames]
            n = int(sum(sizes))
            off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
            cap = max(max(sizes), 1) * max(M, 1)          # a point may be in every mask: room for all lists of a frame
            Btot = int(sum(len(b) for b in boxes))
            pts = torch.from_numpy(np.concatenate(frames).astype(np.float32).reshape(-1, 4)).to(dev) if n else torch.zeros((1, 4), dtype=torch.float32, device=dev)
            o = dict(uv=torch.empty((max(n, 1), 2), dtype=torch.int32, device=dev), label_bits=torch.empty(max(n, 1), dtype=torch.int32, device=dev),
                     valid_idx=torch.empty(max(n, 1), dtype=torch.int64, device=dev), inst_idx=torch.empty((F, cap), dtype=torch.int64, device=dev),
                     count_mb=torch.zeros(max(M * Btot, 1), dtype=torch.int32, device=dev),
                     summary=torch.zeros(F * SUMMARY_DTYPE.itemsize, dtype=torch.uint8, device=dev),
                     uv_valid=torch.empty((max(n, 1), 2), dtype=torch.int32, device=dev),
                     label_valid=torch.empty(max(n, 1), dtype=torch.int32, device=dev))
            mt = torch.from_numpy(np.stack(masks)).to(dev) if M else None
            torch.cuda.synchronize(dev)
            ctx.set_camera(T, K, W, H, 0.0, dmax)
            if M:
                if "+rects" in mode:
                    ctx.set_mask_rects(LpfContext.mask_rects(np.stack(masks)))
                ctx.set_masks(mt, lend="+lent" in mode)
            else:
                ctx.clear_masks()
            ctx.set_boxes(boxes, oriented=oriented)
            ctx.run_device(pts, off, inst_cap=cap, **o)
            held.append((o, pts, mt, off, sizes))
        ctx.sync()
    for (T, K, W, H, dmax, oriented, M, frames, masks, boxes), (o, pts, mt, off, sizes) in zip(cases, held):
        sm = np.frombuffer(o["summary"].cpu().numpy().tobytes(), SUMMARY_DTYPE)
        uv, lab = o["uv"].cpu().numpy(), o["label_bits"].cpu().numpy().view(np.uint32)
        vidx, iidx, cmb = o["valid_idx"].cpu().numpy(), o["inst_idx"].cpu().numpy(), o["count_mb"].cpu().numpy()
        uvv, labv = o["uv_valid"].cpu().numpy(), o["label_valid"].cpu().numpy().view(np.uint32)
        boff = 0
        for f in range(len(frames)):
            a, b = int(off[f]), int(off[f + 1])
            labimg = orc.pack_masks(masks[f], 0, H, W) if M else None
            ref = orc.run(frames[f], T, K, W, H, 0.0, dmax, label_img=labimg, M=M, corners=boxes[f], oriented=oriented, want_float=False)
            assert np.array_equal(uv[a:b, 0], ref["u"]) and np.array_equal(uv[a:b, 1], ref["v"]), (seed, f)
            assert np.array_equal(lab[a:b], ref["label_bits"]), (seed, f)
            assert int(sm[f]["n_valid"]) == ref["n_valid"] and np.array_equal(vidx[a:a + ref["n_valid"]], ref["valid_idx"]), (seed, f)
            nv = ref["n_valid"]
            assert np.array_equal(uvv[a:a + nv, 0], ref["u"][ref["valid_idx"]]) and np.array_equal(uvv[a:a + nv, 1], ref["v"][ref["valid_idx"]])
            assert np.array_equal(labv[a:a + nv], ref["label_bits"][ref["valid_idx"]]), (seed, f)
            assert np.array_equal(sm[f]["inst_count"][:M], ref["inst_count"]), (seed, f)
            for m in range(M):
                lo, hi = int(sm[f]["inst_off"][m]), int(sm[f]["inst_off"][m + 1])
                assert np.array_equal(iidx[f, lo:hi], ref["inst_lists"][m]), (seed, f, m)
            B = len(boxes[f])
            if M and B:
                assert np.array_equal(cmb[M * boff:M * (boff + B)].reshape(M, B), ref["count_mb"]), (seed, f)
            assert np.array_equal(sm[f]["best_box"][:M], ref["best_box"]) and np.array_equal(sm[f]["best_cnt"][:M], ref["best_cnt"]), (seed, f)
            boff += B
