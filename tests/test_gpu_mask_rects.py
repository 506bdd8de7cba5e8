"""lpf_set_mask_rects: a hint -- "mask m of frame f is zero outside this rectangle" -- under which a mask is READ AS ZERO outside its
rectangle, pixel for pixel, by whichever form a launch takes: tiles that read the masks themselves inside the rectangles (launches of
any size: no pack, no label image) or a pack that skips what lies outside.  With rectangles that hold, every output equals the run
without them (and the CPU oracle); with rectangles that do NOT hold, the result is exactly that of masks zeroed outside their
rectangles -- which proves the gating is really done, with the right edges, and that the forms agree."""
import numpy as np
import pytest

from oracle import cpu_oracle as orc

pytestmark = pytest.mark.gpu


def _outside_zeroed(masks, rects):
    """masks [F,M,H,W] with every pixel outside its mask's rectangle (x0, y0, x1, y1; half open) set to zero"""
    out = np.zeros_like(masks)
    F, M = masks.shape[:2]
    for f in range(F):
        for m in range(M):
            x0, y0, x1, y1 = (int(v) for v in rects[f, m])
            if x1 > x0 and y1 > y0:
                out[f, m, max(y0, 0):y1, max(x0, 0):x1] = masks[f, m, max(y0, 0):y1, max(x0, 0):x1]
    return out


def _run(ctx, torch, dev, pts_list, masks, rects, rects_on_device, T, K, W, H, dmax):
    F, M = masks.shape[:2]
    d_masks = torch.from_numpy(masks).to(dev)
    if rects is not None:
        ctx.set_mask_rects(torch.from_numpy(rects).to(dev) if rects_on_device else rects)
    ctx.set_masks(d_masks, lend=True)
    res = ctx.run_batch(pts_list, want_float=False)
    ctx.sync()
    return res


@pytest.mark.parametrize("mode", [False, "fused", "fused-pack"])
@pytest.mark.parametrize("rects_on_device", [False, True])
def test_rectangles_that_hold_change_nothing(calib, mode, rects_on_device):
    import torch
    from lidar_object_detection_amd import synthetic as S
    from lidar_object_detection_amd._native import LpfContext
    _, T, K, W, H = S.default_calibration(calib)
    dev = torch.device("cuda", 0)
    F, M, n = 2, 6, 600_000                                 # (dense frames: the masks are packed, not read by the tiles themselves)
    scs = [S.scene(n, n_masks=M, n_boxes=8, seed=8100 + f, calib=calib) for f in range(F)]
    masks = np.stack([sc["masks"] for sc in scs])
    masks[1, 2] = 0                                         # an empty mask: rectangle 0, 0, 0, 0
    rects = LpfContext.mask_rects(masks)
    assert rects.shape == (F, M, 4) and (rects[1, 2] == 0).all() and (rects[0, 0, 2] > rects[0, 0, 0])
    with LpfContext(0) as ctx:
        ctx.set_pipelined(mode)
        ctx.set_camera(T, K, W, H, 0.0, 40.0)
        ctx.set_boxes([sc["corners_velo"] for sc in scs])
        for k in range(3):                                  # (pipelined modes: the hint travels with its run)
            res = _run(ctx, torch, dev, [sc["points"] for sc in scs], masks, rects if k != 1 else None, rects_on_device, T, K, W, H, 40.0)
            for f, (sc, r) in enumerate(zip(scs, res)):
                lab = orc.pack_masks(masks[f], 0, H, W)
                o = orc.run(sc["points"], T, K, W, H, 0.0, 40.0, label_img=lab, M=M, corners=sc["corners_velo"], want_float=False)
                assert np.array_equal(r["label_bits"], o["label_bits"]), (mode, k, f)
                assert np.array_equal(r["inst_count"], o["inst_count"]) and np.array_equal(r["count_mb"], o["count_mb"]), (mode, k, f)
                for a, b in zip(r["inst_lists"], o["inst_lists"]):
                    assert np.array_equal(a, b)


@pytest.mark.parametrize("mode", [False, "fused-pack"])
@pytest.mark.parametrize("W,H", [(64, 24), (72, 32)])      # 72: the pack's groups of 16 pixels run over the ends of the rows
def test_rectangles_that_do_not_hold_show_what_is_skipped(mode, W, H):
    """host-memory runs: in order the tiles read the lent masks inside the rectangles; a pipelined context packs them (with the
    rectangles) before a host-memory run -- the same result either way"""
    import torch
    from lidar_object_detection_amd._native import LpfContext
    rng = np.random.default_rng(5 + W)
    dev = torch.device("cuda", 0)
    T = np.eye(4)
    K = np.array([[W / 2.0, 0, W / 2.0], [0, H / 2.0, H / 2.0], [0, 0, 1.0]])
    F, M, n = 3, 5, 40_000                                   # 40 000 points on 64 x 24 pixels: dense
    pts_list = []
    for f in range(F):
        z = rng.uniform(2.0, 30.0, n)
        pts_list.append(np.stack([rng.uniform(-1.1, 1.1, n) * z, rng.uniform(-1.1, 1.1, n) * z, z, np.zeros(n)], 1).astype(np.float32))
    masks = (rng.random((F, M, H, W)) < 0.5).astype(np.uint8)               # noise everywhere: the rectangles below do not hold
    rects = np.zeros((F, M, 4), np.int32)
    for f in range(F):
        for m in range(M):
            x0, y0 = rng.integers(0, W - 4), rng.integers(0, H - 2)
            rects[f, m] = (x0, y0, rng.integers(x0 + 1, W + 1), rng.integers(y0 + 1, H + 1))
    rects[0, 0] = (0, 0, W, H); rects[0, 1] = (0, 0, 0, 0); rects[0, 2] = (15, 3, 16, 4); rects[0, 3] = (16, 0, 17, H)
    expect = _outside_zeroed(masks, rects)
    assert (expect != masks).any()
    with LpfContext(0) as ctx:
        ctx.set_pipelined(mode)
        ctx.set_camera(T, K, W, H, 0.0, 50.0)
        for k in range(2):
            res = _run(ctx, torch, dev, pts_list, masks, rects, k == 1, T, K, W, H, 50.0)
            for f, r in enumerate(res):
                o = orc.run(pts_list[f], T, K, W, H, 0.0, 50.0, label_img=orc.pack_masks(expect[f], 0, H, W), M=M, want_float=False)
                assert o["inst_count"].sum() > 1000
                assert np.array_equal(r["label_bits"], o["label_bits"]), (mode, W, k, f)
        # the hint is consumed by the masks that follow it: the next masks are read in full
        res = _run(ctx, torch, dev, pts_list, masks, None, False, T, K, W, H, 50.0)
        o = orc.run(pts_list[0], T, K, W, H, 0.0, 50.0, label_img=orc.pack_masks(masks[0], 0, H, W), M=M, want_float=False)
        assert np.array_equal(res[0]["label_bits"], o["label_bits"])


@pytest.mark.parametrize("mode", [False, "fused", "fused-pack"])
@pytest.mark.parametrize("kind", ["u8", "f32"])
@pytest.mark.parametrize("size", ["large", "dense", "small"])
def test_rectangles_that_do_not_hold_in_device_mode_launches_of_any_size(calib, mode, kind, size):
    """Device-mode steps (the software-pipelined launches included): with the hint, the tiles of a launch of sparse frames of ANY size
    ("large": 20 frames, 4 M points, the large geometry; "small") read the lent masks inside the rectangles only -- no pack rides, no
    label image is written; dense frames ("dense": 900 k points on 530 k pixels) keep the pack, which honours the hint the same way.
    Noise masks and rectangles that do not hold: every frame's labels, lists and counts are those of the masks zeroed outside their
    rectangles, whichever form ran.  uint8 masks and float masks (rule astype)."""
    import torch
    from lidar_object_detection_amd import synthetic as S
    from lidar_object_detection_amd._native import LpfContext, SUMMARY_DTYPE
    _, T, K, W, H = S.default_calibration(calib)
    dev = torch.device("cuda", 0)
    F, M = {"large": (20, 7), "dense": (5, 7), "small": (2, 7)}[size]
    n = {"large": 200_000, "dense": 900_000, "small": 60_000}[size]     # large / dense: > 3.5 Mi points, the large geometry
    rng = np.random.default_rng(91 + F)
    scs = [S.scene(n + 977 * f, n_masks=M, n_boxes=6, seed=8400 + f, calib=calib) for f in range(F)]     # (frames that do not start on a multiple of 64)
    masks = (rng.random((F, M, H, W)) < 0.35).astype(np.uint8)
    rects = np.zeros((F, M, 4), np.int32)
    for f in range(F):
        for m in range(M):
            x0, y0 = rng.integers(0, W - 40), rng.integers(0, H - 20)
            rects[f, m] = (x0, y0, rng.integers(x0 + 1, W + 1), rng.integers(y0 + 1, H + 1))
    rects[0, 0] = (0, 0, W, H); rects[0, 1] = (0, 0, 0, 0); rects[0, 2] = (15, 3, 16, 4); rects[1, 0] = (5, 5, 4, 9)      # whole image, empty, one pixel, inverted
    rects[1, 1] = rects[1, 2] = rects[1, 3] = (300, 50, 900, 300)                                                         # three rectangles on the same pixels
    expect = _outside_zeroed(masks, rects)
    sizes = [len(sc["points"]) for sc in scs]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    ntot, cap = int(off[-1]), max(sizes)
    boff = np.arange(F + 1, dtype=np.int32) * 6
    d_pts = torch.from_numpy(np.concatenate([sc["points"] for sc in scs])).to(dev)
    d_masks = torch.from_numpy(masks if kind == "u8" else masks.astype(np.float32)).to(dev)
    d_rects = torch.from_numpy(rects).to(dev)
    o = dict(uv=torch.empty((ntot, 2), dtype=torch.int32, device=dev), label_bits=torch.empty(ntot, dtype=torch.int32, device=dev),
             valid_idx=torch.empty(ntot, dtype=torch.int64, device=dev), inst_idx=torch.empty((F, cap), dtype=torch.int64, device=dev),
             count_mb=torch.zeros(M * F * 6, dtype=torch.int32, device=dev), summary=torch.zeros(F * SUMMARY_DTYPE.itemsize, dtype=torch.uint8, device=dev))
    torch.cuda.synchronize(dev)
    with LpfContext(0) as ctx:
        ctx.set_pipelined(mode)
        ctx.set_camera(T, K, W, H, 0.0, 40.0)
        ctx.set_boxes([sc["corners_velo"] for sc in scs])
        for k in range(3):                                  # (three runs: the pipelined modes' launches carry the roles of different runs)
            if k == 1:
                ctx.set_mask_rects(rects)                   # host memory: copied into the run's scratch set
            else:
                ctx.set_mask_rects(d_rects)
            ctx.set_masks(d_masks, lend=True)
            ctx.run_device(d_pts, off, inst_cap=cap, **o)
            if k == 0:
                ctx.stats(reset=True)                       # (the first run allocates)
        st = ctx.stats()
        ctx.sync()
    sm = np.frombuffer(o["summary"].cpu().numpy().tobytes(), SUMMARY_DTYPE)
    lab, vidx, inst, cmb = o["label_bits"].cpu().numpy().view(np.uint32), o["valid_idx"].cpu().numpy(), o["inst_idx"].cpu().numpy(), o["count_mb"].cpu().numpy()
    for f, sc in enumerate(scs):
        ref = orc.run(sc["points"], T, K, W, H, 0.0, 40.0, label_img=orc.pack_masks(expect[f], 0, H, W), M=M, corners=sc["corners_velo"], want_float=False)
        a = int(off[f])
        assert ref["inst_count"].sum() > 100 or f == 1
        assert np.array_equal(lab[a:a + sizes[f]], ref["label_bits"]), (mode, kind, f)
        assert np.array_equal(vidx[a:a + ref["n_valid"]], ref["valid_idx"]), (mode, kind, f)
        assert np.array_equal(sm[f]["inst_count"][:M], ref["inst_count"]), (mode, kind, f)
        for m in range(M):
            lo, hi = int(sm[f]["inst_off"][m]), int(sm[f]["inst_off"][m + 1])
            assert np.array_equal(inst[f, lo:hi], ref["inst_lists"][m]), (mode, kind, f, m)
        assert np.array_equal(cmb[M * 6 * f:M * 6 * (f + 1)].reshape(M, 6), ref["count_mb"]), (mode, kind, f)
    if mode:
        assert st["host_waits"] == 0, st


def test_hint_is_ignored_where_masks_are_not_packed_as_they_are(calib):
    """erosion, float masks under the 0.5 rule, another shape: the rectangles (which do not hold here) change nothing"""
    import torch
    from lidar_object_detection_amd import synthetic as S
    from lidar_object_detection_amd._native import LpfContext
    _, T, K, W, H = S.default_calibration(calib)
    dev = torch.device("cuda", 0)
    sc = S.scene(200_000, n_masks=4, n_boxes=4, seed=8200, calib=calib)
    bad = np.tile(np.array([[10, 10, 20, 20]], np.int32), (4, 1))
    with LpfContext(0) as ctx:
        ctx.set_camera(T, K, W, H, 0.0, 40.0)
        ref = {}
        for hint in (False, True):
            out = []
            if hint:
                ctx.set_mask_rects(bad)
            ctx.set_masks(torch.from_numpy(sc["masks"]).to(dev), erode_iters=1)
            out.append(ctx.run(sc["points"], want_float=False)["label_bits"])
            if hint:
                ctx.set_mask_rects(bad)
            ctx.set_masks(torch.from_numpy(sc["masks"].astype(np.float32)).to(dev), binarize="gt0.5")
            out.append(ctx.run(sc["points"], want_float=False)["label_bits"])
            if hint:
                ctx.set_mask_rects(np.tile(bad[None], (2, 1, 1)))          # F = 2: not these masks' shape
            ctx.set_masks(torch.from_numpy(sc["masks"]).to(dev))
            out.append(ctx.run(sc["points"], want_float=False)["label_bits"])
            ref[hint] = out
        for a, b in zip(ref[False], ref[True]):
            assert np.array_equal(a, b) and a.any()


@pytest.mark.parametrize("mode", [False, "fused-pack"])
@pytest.mark.parametrize("cam", ["kitti_1242x375", "narrow_1000x37", "tiny_17x16"])
def test_cameras_whose_size_is_no_multiple_of_the_grid_cell_and_rectangles_beyond_the_image(mode, cam):
    """The candidate grid of the rectangle-reading tiles has 16 x 16-pixel cells and the pack works in groups of 16 pixels: cameras whose
    width / height are no multiple of 16 (KITTI's 1242 x 375; a 37-row strip; 17 x 16, one cell and one pixel), and rectangles that
    start left of / above the image or end beyond it (read as clipped to the image), in a LARGE launch of sparse frames (the grid form)
    and a small one (all mask bytes gated by the rectangles)."""
    import torch
    from lidar_object_detection_amd import synthetic as S
    from lidar_object_detection_amd._native import LpfContext, SUMMARY_DTYPE
    W, H = {"kitti_1242x375": (1242, 375), "narrow_1000x37": (1000, 37), "tiny_17x16": (17, 16)}[cam]
    _, T, K0, W0, H0 = S.default_calibration(None)
    K = np.array(K0, dtype=np.float64).copy()
    K[0, 0] *= W / W0; K[0, 2] *= W / W0; K[1, 1] *= H / H0; K[1, 2] *= H / H0           # the same field of view on the other sensor
    dev = torch.device("cuda", 0)
    M = 6
    for F, n in ((18, 230_000), (2, 50_000)):                                          # > 3.5 Mi points: the large geometry; a small launch
        rng = np.random.default_rng(W * 31 + F)
        clouds = [S.synthetic_cloud(n + 131 * f, seed=9100 + f) for f in range(F)]
        masks = (rng.random((F, M, H, W)) < 0.4).astype(np.uint8)
        rects = np.zeros((F, M, 4), np.int32)
        for f in range(F):
            for m in range(M):
                x0, y0 = rng.integers(-20, W - 1), rng.integers(-20, H - 1)
                rects[f, m] = (x0, y0, rng.integers(x0 + 1, W + 30), rng.integers(y0 + 1, H + 30))
        rects[0, 0] = (-7, -3, W + 9, H + 4); rects[0, 1] = (W - 1, H - 1, W + 100, H + 100); rects[0, 2] = (-50, -50, 1, 1)
        rects[0, 3] = (W, 0, W + 5, H); rects[0, 4] = (0, H, W, H + 5); rects[0, 5] = (-9, -9, 0, 0)      # (the last three: wholly outside)
        big = np.iinfo(np.int32)
        rects[1, 0] = (big.min, big.min, big.max, big.max)                 # "no limit" sentinels: the whole image,
        rects[1, 1] = (W - 3, big.min, big.max, big.max)                   # ... its last three columns,
        rects[1, 2] = (big.min, H - 2, big.max, big.max)                   # ... its last two rows
        clipped = rects.copy()
        clipped[..., 0] = np.clip(rects[..., 0], 0, W); clipped[..., 2] = np.clip(rects[..., 2], 0, W)
        clipped[..., 1] = np.clip(rects[..., 1], 0, H); clipped[..., 3] = np.clip(rects[..., 3], 0, H)
        expect = _outside_zeroed(masks, clipped)
        assert expect[0, 0].any() and not expect[0, 3].any() and not expect[0, 4].any() and not expect[0, 5].any()
        sizes = [len(c) for c in clouds]
        off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        ntot, cap = int(off[-1]), max(sizes)
        d_pts = torch.from_numpy(np.concatenate(clouds)).to(dev)
        d_masks, d_rects = torch.from_numpy(masks).to(dev), torch.from_numpy(rects).to(dev)
        o = dict(uv=torch.empty((ntot, 2), dtype=torch.int32, device=dev), label_bits=torch.empty(ntot, dtype=torch.int32, device=dev),
                 valid_idx=torch.empty(ntot, dtype=torch.int64, device=dev), inst_idx=torch.empty((F, cap), dtype=torch.int64, device=dev),
                 summary=torch.zeros(F * SUMMARY_DTYPE.itemsize, dtype=torch.uint8, device=dev))
        torch.cuda.synchronize(dev)
        with LpfContext(0) as ctx:
            ctx.set_pipelined(mode)
            ctx.set_camera(T, K, W, H, 0.0, 60.0)
            for _ in range(3):
                ctx.set_mask_rects(d_rects)
                ctx.set_masks(d_masks, lend=True)
                ctx.run_device(d_pts, off, inst_cap=cap, **o)
            ctx.sync()
        sm = np.frombuffer(o["summary"].cpu().numpy().tobytes(), SUMMARY_DTYPE)
        lab, vidx, inst = o["label_bits"].cpu().numpy().view(np.uint32), o["valid_idx"].cpu().numpy(), o["inst_idx"].cpu().numpy()
        seen = 0
        for f in range(F):
            ref = orc.run(clouds[f], T, K, W, H, 0.0, 60.0, label_img=orc.pack_masks(expect[f], 0, H, W), M=M, want_float=False)
            a = int(off[f])
            assert np.array_equal(lab[a:a + sizes[f]], ref["label_bits"]), (cam, mode, F, f)
            assert np.array_equal(vidx[a:a + ref["n_valid"]], ref["valid_idx"]), (cam, mode, F, f)
            assert np.array_equal(sm[f]["inst_count"][:M], ref["inst_count"]), (cam, mode, F, f)
            for m in range(M):
                lo, hi = int(sm[f]["inst_off"][m]), int(sm[f]["inst_off"][m + 1])
                assert np.array_equal(inst[f, lo:hi], ref["inst_lists"][m]), (cam, mode, F, f, m)
            seen += int(ref["inst_count"].sum())
        assert seen > (100 if W > 100 else 0), (cam, F, seen)
