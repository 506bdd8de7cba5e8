"""Real scans at the headline's size: 146 frames per step (the four full-size golden frames in turn -- sample frames 100, 1461, 2098 and
2449, 16.9 M points, their five masks each and all of their annotated boxes, 31 / 21 / 186 / 314, filtered on the device), depth < 50,
as device-mode steps in order and software-pipelined, with and without the masks' rectangles (lpf_set_mask_rects).  With the
rectangles the tiles of these LARGE launches read the lent masks inside the rectangles themselves -- no pack, no label image.  EVERY
frame of the last step is compared with the reference-generated golden vectors: pixels, labels at the valid points, valid_idx,
EVERY instance list, the per-(mask, box) counts and the best boxes (V3:565-592, V3:211-233, V3:344-379; cvs_erosion.py:86-87, 110
for the rectangles)."""
import numpy as np
import pytest

from conftest import GOLDEN, check_full

pytestmark = pytest.mark.gpu

NFR = 146
NAMES = ("frame_0000000100.npz", "frame_0000001461_full.npz", "frame_0000002098_full.npz", "frame_0000002449_full.npz")
TAG = "rect5_d50"


@pytest.mark.parametrize("mode", ["serial", "fused", "fused-pack"])
@pytest.mark.parametrize("rects", ["rects", "norects"])
def test_every_list_of_every_frame_at_146_real_frames(calib, mode, rects):
    import os
    import torch
    from lidar_object_detection_amd._native import LpfContext, SUMMARY_DTYPE
    W, H = int(calib["width"]), int(calib["height"])
    T, K = np.asarray(calib["TrVeloToRect"]), np.asarray(calib["K"])[:3, :3]
    Tcv = np.linalg.inv(np.asarray(calib["TrVeloToCam"]))
    dev = torch.device("cuda", 0)
    gs = [dict(np.load(os.path.join(GOLDEN, n))) for n in NAMES]
    fr = [dict(points=np.ascontiguousarray(g["points"], dtype=np.float32),
               masks=np.unpackbits(g["masks_rect5_packed"], axis=-1)[..., :W].astype(np.uint8),
               cam0=np.ascontiguousarray(g["corners_cam0_raw"], dtype=np.float64)) for g in gs]
    batch = [i % 4 for i in range(NFR)]
    sizes = [len(fr[i]["points"]) for i in batch]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    nb = [len(fr[i]["cam0"]) for i in batch]
    boff = np.concatenate([[0], np.cumsum(nb)]).astype(np.int32)
    M, ntot, cap = 5, int(off[-1]), max(sizes)
    d_pts = torch.from_numpy(np.concatenate([fr[i]["points"] for i in batch])).to(dev)
    masks = np.stack([fr[i]["masks"] for i in batch])
    d_masks = torch.from_numpy(masks).to(dev)
    d_cam0 = torch.from_numpy(np.concatenate([fr[i]["cam0"] for i in batch])).to(dev)
    d_rects = torch.from_numpy(LpfContext.mask_rects(masks)).to(dev) if rects == "rects" else None
    o = dict(uv=torch.empty((ntot, 2), dtype=torch.int32, device=dev), label_bits=torch.empty(ntot, dtype=torch.int32, device=dev),
             valid_idx=torch.empty(ntot, dtype=torch.int64, device=dev), inst_idx=torch.empty((NFR, cap), dtype=torch.int64, device=dev),
             count_mb=torch.zeros(M * int(boff[-1]), dtype=torch.int32, device=dev), summary=torch.zeros(NFR * SUMMARY_DTYPE.itemsize, dtype=torch.uint8, device=dev))
    torch.cuda.synchronize(dev)
    with LpfContext(0) as ctx:
        ctx.set_pipelined(False if mode == "serial" else mode)
        ctx.set_camera(T, K, W, H, 0.0, 50.0)
        fn = ctx.make_device_step(d_pts, off, masks_u8=d_masks, lend=True, boxes_cam0=d_cam0, box_off=boff, T_cam_to_velo=Tcv, mask_rects=d_rects,
                                  inst_cap=cap, **o)
        for _ in range(2):                                  # (allocations; the pipelined modes' launches then carry the roles of three runs)
            fn()
        ctx.sync()
        for t in o.values():
            t.zero_()
        torch.cuda.synchronize(dev)
        ctx.stats(reset=True)
        for _ in range(3):
            fn()
        st = ctx.stats()
        ctx.sync()
    assert st["host_waits"] == 0 and st["drains"] == 0, st
    sm = np.frombuffer(o["summary"].cpu().numpy().tobytes(), SUMMARY_DTYPE)
    uv, lab, vidx = o["uv"].cpu().numpy(), o["label_bits"].cpu().numpy().view(np.uint32), o["valid_idx"].cpu().numpy()
    inst, cmb = o["inst_idx"].cpu().numpy(), o["count_mb"].cpu().numpy()
    for f in range(NFR):
        g, a, n = gs[batch[f]], int(off[f]), sizes[f]
        why = (mode, rects, f)
        check_full(g, "u", uv[a:a + n, 0], np.int64)
        check_full(g, "v", uv[a:a + n, 1], np.int64)
        nv = int(sm[f]["n_valid"])
        check_full(g, "valid_idx_d50", vidx[a:a + nv], np.int64)
        check_full(g, "bg_assigned_" + TAG, np.packbits(lab[a:a + n][vidx[a:a + nv]] != 0), np.uint8)
        assert int(sm[f]["n_labelled"]) == int(np.count_nonzero(lab[a:a + n])), why
        assert np.array_equal(sm[f]["inst_count"][:M], g["inst_count_" + TAG]), why
        tot = int(sm[f]["inst_off"][M])
        assert tot == int(g["inst_count_" + TAG].sum()) and int(sm[f]["inst_overflow"]) == 0, why
        check_full(g, "inst_cat_" + TAG, inst[f, :tot], np.int64)                     # every list, concatenated in mask order
        for m in range(M):                                                           # ... and each list's labels say so
            lo, hi = int(sm[f]["inst_off"][m]), int(sm[f]["inst_off"][m + 1])
            assert hi - lo == int(g["inst_count_" + TAG][m]) and ((lab[a:a + n][inst[f, lo:hi]] >> m) & 1).all(), why + (m,)
        vis = g["visible_pos"]
        got = cmb[M * int(boff[f]):M * int(boff[f + 1])].reshape(M, nb[f])
        assert np.array_equal(got[:, vis], g["count_mb_" + TAG]), why
        dropped = np.ones(nb[f], bool); dropped[vis] = False
        assert not got[:, dropped].any(), why
        want_cnt = g["count_mb_" + TAG].max(axis=1) if len(vis) else np.zeros(M, np.int64)
        want_box = np.where(want_cnt > 0, vis[g["count_mb_" + TAG].argmax(axis=1)] if len(vis) else -1, -1)
        assert np.array_equal(sm[f]["best_cnt"][:M], want_cnt) and np.array_equal(sm[f]["best_box"][:M], want_box), why
