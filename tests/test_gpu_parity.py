"""HIP path (through the C ABI) against the committed golden vectors and the CPU oracle.

Bar: bit-exact for every integer output (pixels, labels, index lists, counts, best
boxes); |delta| <= 1e-5 (relative to max(1,|x|)) for the pre-rounding floats, which is
the tolerance BASELINE.json's north_star states.
"""
import numpy as np
import pytest

from conftest import check_full, load_golden_full, golden_frames, load_golden, unpack_masks
from oracle import cpu_oracle as orc

pytestmark = pytest.mark.gpu

FRAMES = golden_frames()["frames"]
FS = golden_frames()["float_stride"]
I32 = np.iinfo(np.int32)
TOL = 1e-5


@pytest.fixture(scope="module")
def ctx():
    """One context of the product library for the module: the launch geometry follows the launch size, as it does for users
    (forced geometries: tests/test_gpu_fuzz.py on the lab build; large launches at full size: tests/test_gpu_headline.py)."""
    from lidar_object_detection_amd._native import LpfContext
    c = LpfContext(0)
    yield c
    c.close()


def _close(a, b):
    fin = np.isfinite(b)
    assert np.array_equal(np.isfinite(a), fin)
    if fin.any():
        err = np.abs(a[fin] - b[fin]) / np.maximum(1.0, np.abs(b[fin]))
        assert err.max() <= TOL, err.max()


def _compare(r, o, M, want_float=True):
    """r: HIP result dict, o: oracle result dict."""
    assert np.array_equal(r["u"], o["u"])
    assert np.array_equal(r["v"], o["v"])
    assert np.array_equal(r["label_bits"], o["label_bits"])
    assert r["n_valid"] == o["n_valid"]
    assert np.array_equal(r["valid_idx"], o["valid_idx"])
    assert r["n_labelled"] == int(np.count_nonzero(o["label_bits"]))
    assert np.array_equal(r["inst_count"], o["inst_count"])
    assert len(r["inst_lists"]) == M
    for a, b in zip(r["inst_lists"], o["inst_lists"]):
        assert np.array_equal(a, b)
    assert np.array_equal(r["count_mb"], o["count_mb"])
    if o["count_mb"].shape[1]:
        assert np.array_equal(r["best_box"], o["best_box"])
        assert np.array_equal(r["best_cnt"], o["best_cnt"])
    if want_float:
        for k in ("depth", "uf", "vf"):
            _close(r[k], o[k])                       # the contract: 1e-5
            # observed on gfx950: the kernel's float64 arithmetic is bit-identical to the oracle's
            assert np.array_equal(r[k], o[k], equal_nan=True), k


@pytest.mark.parametrize("rec", FRAMES, ids=lambda r: "f%d" % r["frame"])
@pytest.mark.parametrize("tag", ["rect5_d50", "rect5_d30", "edge_d50"])
def test_golden_frames(ctx, calib, rec, tag):
    g = load_golden(rec["frame"])
    if "u" not in g:
        pytest.skip("frame skipped by the reference (no boxes)")
    kind, dmax = tag.split("_d")
    W, H = int(calib["width"]), int(calib["height"])
    masks = unpack_masks(g, kind, H, W)
    M = masks.shape[0]
    ctx.set_camera(calib["TrVeloToRect"], calib["K"], W, H, 0.0, float(dmax))
    ctx.set_masks(masks)                      # float32 masks, V2/V4 semantics
    ctx.set_boxes(g["corners_velo"], oriented=True)
    r = ctx.run(g["points"], want_float=True)
    # --- against the reference's own outputs ---
    assert np.array_equal(r["u"], np.clip(g["u"], I32.min, I32.max).astype(np.int32))
    assert np.array_equal(r["v"], np.clip(g["v"], I32.min, I32.max).astype(np.int32))
    assert np.array_equal(r["valid_idx"], g["valid_idx_d" + dmax])
    assert np.array_equal(r["inst_count"], g["inst_count_" + tag])
    cat = np.concatenate(r["inst_lists"]) if M else np.zeros(0, np.int64)
    assert np.array_equal(cat, g["inst_cat_" + tag])
    assert np.array_equal(r["count_mb"], g["count_mb_" + tag])
    for k in ("depth", "uf", "vf"):
        _close(r[k][::FS], g[k + "_s"])
    rows = [m for m in range(M) if r["inst_count"][m] > 0] if g["corners_velo"].shape[0] else []
    matched = np.array([r["best_box"][m] if r["best_cnt"][m] >= 10 else -1 for m in rows], np.int64)
    inside = np.array([r["best_cnt"][m] if r["best_cnt"][m] >= 10 else 0 for m in rows], np.int64)
    assert np.array_equal(matched, g["stats_matched_bbox_id_" + tag])
    assert np.array_equal(inside, g["stats_points_inside_bbox_" + tag])
    # --- and against the oracle, every output ---
    lab = orc.pack_masks(orc.binarize_f32(masks, 0), 0, H, W)
    assert np.array_equal(ctx.get_label_image()[0], lab)
    o = orc.run(g["points"], calib["TrVeloToRect"], calib["K"][:, :3], W, H, 0.0, float(dmax),
                label_img=lab, M=M, corners=g["corners_velo"], oriented=True)
    _compare(r, o, M)
    # AABB variant (use_oriented=False, V3:143-164)
    ctx.set_boxes(g["corners_velo"], oriented=False)
    ra = ctx.run(g["points"], want_float=False)
    assert np.array_equal(ra["count_mb"], g["count_mb_aabb_" + tag])


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 1023, 1024, 1025, 4097, 70001])
def test_ragged_sizes_vs_oracle(ctx, calib, n):
    from lidar_object_detection_amd import synthetic as S
    _, T, K, W, H = S.default_calibration(calib)
    sc = S.scene(max(n, 1), n_masks=3, n_boxes=4, seed=n)
    pts = sc["points"][:n]
    ctx.set_camera(T, K, W, H, 0.0, 30.0)
    ctx.set_masks(sc["masks"])
    ctx.set_boxes(sc["corners_velo"])
    r = ctx.run(pts, want_float=True)
    lab = orc.pack_masks(sc["masks"], 0, H, W)
    o = orc.run(pts, T, K, W, H, 0.0, 30.0, label_img=lab, M=3, corners=sc["corners_velo"])
    _compare(r, o, 3)


def test_no_masks_no_boxes(ctx, calib):
    from lidar_object_detection_amd import synthetic as S
    _, T, K, W, H = S.default_calibration(calib)
    pts = S.synthetic_cloud(50000, seed=5)
    ctx.set_camera(T, K, W, H, 0.0, 50.0)
    ctx.clear_masks()
    ctx.clear_boxes()
    r = ctx.run(pts, want_float=True)
    o = orc.run(pts, T, K, W, H, 0.0, 50.0)
    _compare(r, o, 0)
    assert r["n_labelled"] == 0 and not r["label_bits"].any()


def test_m32_overlapping_masks_and_many_boxes(ctx, calib):
    """All 32 label bits in use, heavy overlap (a point in many instances), 200 boxes."""
    from lidar_object_detection_amd import synthetic as S
    _, T, K, W, H = S.default_calibration(calib)
    sc = S.scene(300000, n_masks=32, n_boxes=200, seed=11)
    sc["masks"][31] = 1                                   # full-image mask: every valid point
    ctx.set_camera(T, K, W, H, 0.0, 50.0)
    ctx.set_masks(sc["masks"])
    ctx.set_boxes(sc["corners_velo"])
    r = ctx.run(sc["points"], want_float=False)
    lab = orc.pack_masks(sc["masks"], 0, H, W)
    o = orc.run(sc["points"], T, K, W, H, 0.0, 50.0, label_img=lab, M=32, corners=sc["corners_velo"],
                want_float=False)
    _compare(r, o, 32, want_float=False)
    assert r["inst_count"][31] == r["n_valid"]


def test_more_boxes_than_the_summary_stages(ctx, calib):
    """B = 2100 > LPF_FIN_STAGE (2048): the per-frame summary reads the inside counts from memory; 33 candidate words per cell."""
    from lidar_object_detection_amd import synthetic as S
    _, T, K, W, H = S.default_calibration(calib)
    sc = S.scene(120000, n_masks=3, n_boxes=2100, seed=23)
    ctx.set_camera(T, K, W, H, 0.0, 60.0)
    ctx.set_masks(sc["masks"])
    ctx.set_boxes(sc["corners_velo"])
    lab = orc.pack_masks(sc["masks"], 0, H, W)
    o = orc.run(sc["points"], T, K, W, H, 0.0, 60.0, label_img=lab, M=3, corners=sc["corners_velo"], want_float=False)
    for _ in range(2):                                      # twice: the counters must come back clean
        r = ctx.run(sc["points"], want_float=False)
        _compare(r, o, 3, want_float=False)


@pytest.mark.parametrize("oriented", [True, False])
def test_ground_grid_extremes(ctx, calib, oriented):
    """The candidate structure (a ground grid per 64 boxes, lpf_box_frame_block) only skips hopeless pairs: counts must not depend
    on it.  One word holds 40 annotations of the same car a few centimetres apart (a moving car, one box per timestamp: every cell of
    them holds them all), a box a million metres away (it stretches the domain: the cars fall into one cell), a box 2e30 m long (its
    float bounds leave the floats' range), a degenerate box (zero volume), a box with a NaN corner, a box with an infinite corner;
    a second word holds ordinary boxes.  Points: a cloud, plus points on and just beyond the faces of the first car."""
    from lidar_object_detection_amd import synthetic as S
    TrVeloToCam, T, K, W, H = S.default_calibration(calib)
    rng = np.random.default_rng(77)
    _, car = S.synthetic_boxes(1, seed=5, velo_to_cam=TrVeloToCam)
    stack = np.concatenate([car + rng.uniform(-0.05, 0.05, 3) for _ in range(40)])
    far = car + np.array([1.0e6, -3.0e5, 0.0])
    long_box = car.copy(); long_box[0, :, 0] *= 1.0e30
    flat = car.copy(); flat[0, :, 2] = flat[0, 0, 2]
    nan_box = car.copy(); nan_box[0, 3, 1] = np.nan
    inf_box = car.copy(); inf_box[0, 6, 0] = np.inf
    _, others = S.synthetic_boxes(70, seed=9, velo_to_cam=TrVeloToCam)
    corners = np.concatenate([stack, far, long_box, flat, nan_box, inf_box, others])
    assert 64 < corners.shape[0] <= 128
    sc = S.scene(150000, n_masks=4, n_boxes=1, seed=31)
    lo, hi = car[0].min(0), car[0].max(0)
    extra = [[x, y, z, 0.0] for x in (lo[0], hi[0], np.nextafter(np.float32(lo[0]), np.float32(-1e9)), 0.5 * (lo[0] + hi[0]))
             for y in (lo[1], hi[1], 0.5 * (lo[1] + hi[1])) for z in (lo[2], hi[2], 0.5 * (lo[2] + hi[2]))]
    inside = rng.uniform(lo - 0.2, hi + 0.2, (4000, 3))
    pts = np.concatenate([sc["points"], np.array(extra, np.float32), np.concatenate([inside, np.zeros((4000, 1))], 1).astype(np.float32)])
    masks = sc["masks"].copy()
    masks[3] = 1                                           # every valid point is a masked point
    ctx.set_camera(T, K, W, H, 0.0, 80.0)
    ctx.set_masks(masks)
    ctx.set_boxes(corners, oriented=oriented)
    r = ctx.run(pts, want_float=False)
    lab = orc.pack_masks(masks, 0, H, W)
    o = orc.run(pts, T, K, W, H, 0.0, 80.0, label_img=lab, M=4, corners=corners, oriented=oriented, want_float=False)
    _compare(r, o, 4, want_float=False)
    assert o["count_mb"][3, :40].min() > 100               # the stacked boxes do hold points


def test_constructed_edge_points(ctx):
    """depth == 0, exact .5 rounding ties, pixels W-1/H-1 and W/H, NaN/inf, points on slab faces."""
    T = np.eye(4)
    K = np.array([[2.0, 0, 8.0], [0, 2.0, 4.0], [0, 0, 1.0]])
    W, H = 16, 8
    z = 4.0
    xs = []
    for uq in (-0.5, 0.5, 1.5, 2.5, 14.5, 15.5, 15.0, 15.49, 16.0, -0.49, 0.0):   # ties -> half to even
        xs.append([(uq - 8.0) * z / 2.0, 0.0, z, 0.0])
    for vq in (-0.5, 0.5, 6.5, 7.5, 7.0, 8.0):
        xs.append([0.0, (vq - 4.0) * z / 2.0, z, 0.0])
    xs += [[1.0, 1.0, 0.0, 0.0], [0.0, 0.0, 0.0, 0.0], [1.0, 2.0, -3.0, 0.0], [0.0, 0.0, 50.0, 0.0],
           [0.0, 0.0, np.nextafter(np.float32(50.0), np.float32(0)), 0.0], [0.0, 0.0, 1e-30, 0.0],
           [np.nan, 0.0, 1.0, 0.0], [np.inf, 0.0, 1.0, 0.0], [0.0, -np.inf, 1.0, 0.0], [1e30, 1e30, 1e-3, 0.0],
           [0.0, 0.0, np.inf, 0.0], [3e38, 0.0, 1e-38, 0.0]]
    # points on / next to the faces of a unit-ish box (corner order of the dataset)
    from lidar_object_detection_amd.synthetic import _CORNER_HWL
    c0 = np.array([-1.0, -1.0, 3.0])
    corners = c0 + _CORNER_HWL @ np.diag([2.0, 2.0, 2.0])
    for p in ([-1, -1, 3], [1, 1, 5], [-1, 0, 4], [1, 0, 4], [0, 0, 3], [0, 0, 5],
              [np.nextafter(np.float32(-1), np.float32(-2)), 0, 4], [np.nextafter(np.float32(1), np.float32(2)), 0, 4],
              [0, 0, 4], [0.999, 0.999, 4.999]):
        xs.append([p[0], p[1], p[2], 0.0])
    pts = np.array(xs, np.float32)
    masks = np.ones((2, H, W), np.uint8)
    masks[1, :, : W // 2] = 0
    ctx.set_camera(T, K, W, H, 0.0, 50.0)
    ctx.set_masks(masks)
    ctx.set_boxes(corners[None])
    r = ctx.run(pts, want_float=True)
    lab = orc.pack_masks(masks, 0, H, W)
    o = orc.run(pts, T, K, W, H, 0.0, 50.0, label_img=lab, M=2, corners=corners[None])
    _compare(r, o, 2)
    assert o["n_valid"] > 10 and o["count_mb"].sum() > 0
    # same bits for the floats, including inf / nan placement
    for k in ("depth", "uf", "vf"):
        assert np.array_equal(r[k], o[k], equal_nan=True)


@pytest.mark.parametrize("mode", ["u8", "f32_raw", "f32_v3", "f32_gt"])
@pytest.mark.parametrize("iters", [0, 1, 2, 3])
def test_mask_pack_and_erosion(ctx, calib, mode, iters):
    rng = np.random.default_rng(3)
    W, H = int(calib["width"]), int(calib["height"])
    M = 7
    base = (rng.random((M, H, W)) < 0.9).astype(np.uint8)
    base[0] = 1                     # full mask: only the border rule matters
    base[1] = 0
    base[2, 100:200, 300:900] = 1
    ctx.set_camera(calib["TrVeloToRect"], calib["K"], W, H, 0.0, 50.0)
    if mode == "u8":
        m = base * np.uint8(255)
        ctx.set_masks(m, erode_iters=iters)
        member = base
    else:
        m = base.astype(np.float32)
        m[3] *= rng.choice(np.array([0.0, 0.5, 0.50000006, 0.999, 1.0, 1.5, 2.0, 256.0, np.nan, -1.0], np.float32), size=(H, W))
        b = {"f32_raw": "astype", "f32_v3": "v3", "f32_gt": "gt0.5"}[mode]
        ctx.set_masks(m, erode_iters=iters, binarize=b)
        member = orc.binarize_f32(m, {"astype": 0, "v3": 1, "gt0.5": 2}[b])
    want = orc.pack_masks(member, iters, H, W)
    assert np.array_equal(ctx.get_label_image()[0], want)


@pytest.mark.parametrize("binarize", ["astype", "v3", "gt0.5"])
@pytest.mark.parametrize("iters", [0, 1, 2])
def test_float_masks_from_a_device_tensor(ctx, calib, binarize, iters):
    """SURVEY 8f-2's headline path: ``result.masks.data`` stays on the GPU (a float32 torch tensor, V3:72 minus the
    .cpu().numpy()); threshold, erosion and packing happen there, for each of the three ways the reference reads a float mask."""
    import torch
    rng = np.random.default_rng(11)
    W, H = int(calib["width"]), int(calib["height"])
    F, M = 2, 6
    base = (rng.random((F, M, H, W)) < 0.85).astype(np.float32)
    base[0, 0] = 1.0
    base[1, 1] = 0.0
    base[0, 3] *= rng.choice(np.array([0.0, 0.5, 0.50000006, 0.999, 1.0, 1.5, 2.0, 256.0, np.nan, -1.0], np.float32), size=(H, W))
    base[1, 4] *= rng.random((H, W)).astype(np.float32)                       # sigmoid-like values, not only 0 / 1
    ctx.set_camera(calib["TrVeloToRect"], calib["K"], W, H, 0.0, 50.0)
    dev = torch.device("cuda", 0)
    t = torch.from_numpy(base).to(dev)
    torch.cuda.synchronize(dev)                                           # (the fixture's context runs on its own stream)
    ctx.set_masks(t, erode_iters=iters, binarize=binarize)
    got = ctx.get_label_image()
    for f in range(F):
        member = orc.binarize_f32(base[f], {"astype": 0, "v3": 1, "gt0.5": 2}[binarize])
        assert np.array_equal(got[f], orc.pack_masks(member, iters, H, W)), f
    # ... and the hot path on those label images
    from lidar_object_detection_amd import synthetic as S
    _, T, K, _, _ = S.default_calibration(calib)
    ctx.set_camera(T, K, W, H, 0.0, 50.0)
    ctx.set_masks(t, erode_iters=iters, binarize=binarize)
    pts = S.synthetic_cloud(60_000, seed=5)
    ctx.clear_boxes()
    rs = ctx.run_batch([pts, pts[:30_000]])
    for f, (r, p) in enumerate(zip(rs, (pts, pts[:30_000]))):
        member = orc.binarize_f32(base[f], {"astype": 0, "v3": 1, "gt0.5": 2}[binarize])
        o = orc.run(p, T, K, W, H, 0.0, 50.0, label_img=orc.pack_masks(member, iters, H, W), M=M, corners=None, want_float=False)
        assert np.array_equal(r["label_bits"], o["label_bits"]) and np.array_equal(r["inst_count"], o["inst_count"])
        assert int(o["inst_count"].sum()) > 0


@pytest.mark.parametrize("kind", ["u8", "f32_raw", "f32_v3", "f32_gt"])
@pytest.mark.parametrize("where", ["host", "lent"])
@pytest.mark.parametrize("M", [3, 12, 32])
def test_unpacked_masks_are_looked_up_directly(ctx, calib, kind, where, M):
    """Host masks and device masks lent to the context (on_device = 2) that need no erosion are not packed when they are set:
    a small launch looks the M mask values of a valid point up directly (LpfDirect in K1), a large one packs them first.  Every
    membership rule, M beyond one round of eight gathers, two runs on the same masks, and the label image read back after."""
    import torch
    from lidar_object_detection_amd import synthetic as S
    rng = np.random.default_rng(100 + M)
    W, H = int(calib["width"]), int(calib["height"])
    _, T, K, _, _ = S.default_calibration(calib)
    F = 2
    base = (rng.random((F, M, H, W)) < 0.3).astype(np.uint8)
    base[0, 0] = 1
    base[1, 1] = 0
    rule = {"u8": None, "f32_raw": "astype", "f32_v3": "v3", "f32_gt": "gt0.5"}[kind]
    if kind == "u8":
        m = base * np.uint8(200)
        member = base
    else:
        m = base.astype(np.float32)
        m[0, 2] *= rng.choice(np.array([0.0, 0.5, 0.50000006, 0.999, 1.0, 1.5, 2.0, 256.0, np.nan, -1.0], np.float32), size=(H, W))
        m[1, 0] *= rng.random((H, W)).astype(np.float32)
        member = np.stack([orc.binarize_f32(m[f], {"astype": 0, "v3": 1, "gt0.5": 2}[rule]) for f in range(F)])
    ctx.set_camera(T, K, W, H, 0.0, 50.0)
    kw = {} if rule is None else {"binarize": rule}
    if where == "host":
        ctx.set_masks(m, **kw)
    else:
        t = torch.from_numpy(m).to(torch.device("cuda", 0))
        torch.cuda.synchronize()
        ctx.set_masks(t, lend=True, **kw)
    clouds = [S.synthetic_cloud(40_000, seed=7), S.synthetic_cloud(25_001, seed=8)]
    boxes = [S.synthetic_boxes(9, seed=3)[1], S.synthetic_boxes(70, seed=4)[1]]
    ctx.set_boxes(boxes)
    want = [orc.run(clouds[f], T, K, W, H, 0.0, 50.0, label_img=orc.pack_masks(member[f], 0, H, W), M=M, corners=boxes[f], want_float=False)
            for f in range(F)]
    for rep in range(2):                                    # the masks are set once
        for f, r in enumerate(ctx.run_batch(clouds)):
            o = want[f]
            for k in ("u", "v", "label_bits", "valid_idx", "count_mb", "best_box", "inst_count"):
                assert np.array_equal(r[k], o[k]), (rep, f, k)
            for a, b in zip(r["inst_lists"], o["inst_lists"]):
                assert np.array_equal(a, b)
    got = ctx.get_label_image()                             # packs them now
    for f in range(F):
        assert np.array_equal(got[f], orc.pack_masks(member[f], 0, H, W))
    r = ctx.run_batch(clouds)                               # ... and the packed image serves the next run
    assert all(np.array_equal(r[f]["label_bits"], want[f]["label_bits"]) for f in range(F))
    ctx.clear_boxes()


def test_batch_equals_single_frames(ctx, calib):
    """run_batch over ragged frames == frame-by-frame runs == oracle."""
    from lidar_object_detection_amd import synthetic as S
    _, T, K, W, H = S.default_calibration(calib)
    sizes = [5000, 0, 1, 123457, 1024, 40000]
    scenes = [S.scene(max(n, 1), n_masks=4, n_boxes=3 + i, seed=20 + i) for i, n in enumerate(sizes)]
    frames = [sc["points"][:n] for sc, n in zip(scenes, sizes)]
    masks = np.stack([sc["masks"] for sc in scenes])
    ctx.set_camera(T, K, W, H, 0.0, 30.0)
    ctx.set_masks(masks)
    ctx.set_boxes([sc["corners_velo"] for sc in scenes])
    rs = ctx.run_batch(frames, want_float=True, want_valid_uv=True)
    lean = ctx.run_batch(frames, want_uv=False, want_label=False, want_valid_uv=True)     # dense arrays not requested
    for f, (sc, p) in enumerate(zip(scenes, frames)):
        lab = orc.pack_masks(sc["masks"], 0, H, W)
        o = orc.run(p, T, K, W, H, 0.0, 30.0, label_img=lab, M=4, corners=sc["corners_velo"])
        _compare(rs[f], o, 4)
        for r in (rs[f], lean[f]):                          # compact outputs == u[valid], v[valid], labels[valid] (V3:590-591)
            vi = o["valid_idx"]
            assert np.array_equal(r["valid_idx"], vi)
            assert np.array_equal(r["u_valid"], o["u"][vi]) and np.array_equal(r["v_valid"], o["v"][vi])
            assert np.array_equal(r["label_valid"], o["label_bits"][vi])
        assert "u" not in lean[f] and "label_bits" not in lean[f]


def test_inst_capacity_overflow_is_reported_and_recovered(ctx, calib):
    from lidar_object_detection_amd import synthetic as S
    _, T, K, W, H = S.default_calibration(calib)
    sc = S.scene(200000, n_masks=6, n_boxes=2, seed=31)
    sc["masks"][:] = 1                                   # every valid point in all 6 instances
    ctx.set_camera(T, K, W, H, 0.0, 50.0)
    ctx.set_masks(sc["masks"])
    ctx.set_boxes(sc["corners_velo"])
    r = ctx.run(sc["points"], inst_cap=16)               # far too small: wrapper re-runs with the exact size
    lab = orc.pack_masks(sc["masks"], 0, H, W)
    o = orc.run(sc["points"], T, K, W, H, 0.0, 50.0, label_img=lab, M=6, corners=sc["corners_velo"], want_float=False)
    _compare(r, o, 6, want_float=False)


def test_repeated_runs_are_identical(ctx, calib):
    """Self-cleaning counters: the same call twice gives the same integers."""
    from lidar_object_detection_amd import synthetic as S
    _, T, K, W, H = S.default_calibration(calib)
    sc = S.scene(150000, seed=41)
    ctx.set_camera(T, K, W, H, 0.0, 30.0)
    ctx.set_masks(sc["masks"])
    ctx.set_boxes(sc["corners_velo"])
    a = ctx.run(sc["points"])
    b = ctx.run(sc["points"])
    for k in ("u", "v", "label_bits", "valid_idx", "count_mb", "best_box", "best_cnt", "inst_count"):
        assert np.array_equal(a[k], b[k]), k


def test_full_size_properties_2m(ctx, calib):
    """BASELINE configs[2] size (2 M points, 8 masks, 32 boxes): size-independent properties
    + full comparison with the oracle (the C oracle does 2 M points in well under a second)."""
    from lidar_object_detection_amd import synthetic as S
    _, T, K, W, H = S.default_calibration(calib)
    sc = S.scene(2_000_000, n_masks=8, n_boxes=32, seed=0)
    ctx.set_camera(T, K, W, H, 0.0, 30.0)
    ctx.set_masks(sc["masks"])
    ctx.set_boxes(sc["corners_velo"])
    r = ctx.run(sc["points"], want_float=False)
    vi = r["valid_idx"]
    assert np.all(np.diff(vi) > 0)                                       # sorted, unique
    assert np.all((r["u"][vi] >= 0) & (r["u"][vi] < W) & (r["v"][vi] >= 0) & (r["v"][vi] < H))
    assert not r["label_bits"][np.setdiff1d(np.arange(len(sc["points"])), vi)].any()
    for m, lst in enumerate(r["inst_lists"]):
        assert np.all(np.diff(lst) > 0)
        assert np.array_equal(lst, np.nonzero((r["label_bits"] >> m) & 1)[0])
        assert np.all(r["count_mb"][m] <= len(lst))
    lab = orc.pack_masks(sc["masks"], 0, H, W)
    o = orc.run(sc["points"], T, K, W, H, 0.0, 30.0, label_img=lab, M=8, corners=sc["corners_velo"], want_float=False)
    _compare(r, o, 8, want_float=False)


@pytest.mark.parametrize("pipelined", [False, "fused", "fused+lent", "fused+lent+large", "fused-pack", "fused-pack+lent", "fused-pack+lent+large"])
def test_device_mode_back_to_back_runs(calib, pipelined):
    """Device-pointer mode (torch tensors): several different batches enqueued back to back without host syncs, in order and
    in the software-pipelined modes (masks packed at the call, or lent; "+large": the large launch geometry forced on the lab
    build, where mode 4 lets the pack ride and mode 2 packs by a launch of its own); every run must equal the oracle."""
    import torch
    from conftest import context_for_form
    from lidar_object_detection_amd import synthetic as S
    from lidar_object_detection_amd._native import SUMMARY_DTYPE
    _, T, K, W, H = S.default_calibration(calib)
    dev = torch.device("cuda", 0)
    fmode = pipelined.split("+")[0] if isinstance(pipelined, str) and pipelined.startswith("fused") else None
    lent = isinstance(pipelined, str) and "+lent" in pipelined
    ctx = context_for_form("large" if (fmode and pipelined.endswith("+large")) else "auto")
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    ctx.set_pipelined(fmode if fmode else False)
    ctx.set_camera(T, K, W, H, 0.0, 30.0)
    F, M, Bx = 3, 5, 7
    runs = []
    for k in range(6):
        sizes = [30000 + 4097 * k, 1 + k, 70000 - 5000 * k]
        scenes = [S.scene(n, n_masks=M, n_boxes=Bx, seed=100 * k + f) for f, n in enumerate(sizes)]
        off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        pts = torch.from_numpy(np.concatenate([sc["points"] for sc in scenes])).to(dev)
        masks = torch.from_numpy(np.stack([sc["masks"] for sc in scenes])).to(dev)
        n = int(off[-1])
        o = dict(uv=torch.empty((n, 2), dtype=torch.int32, device=dev), label_bits=torch.empty(n, dtype=torch.int32, device=dev),
                 valid_idx=torch.empty(n, dtype=torch.int64, device=dev), inst_idx=torch.empty((F, n), dtype=torch.int64, device=dev),
                 count_mb=torch.zeros(F * M * Bx, dtype=torch.int32, device=dev),
                 summary=torch.zeros(F * SUMMARY_DTYPE.itemsize, dtype=torch.uint8, device=dev))
        runs.append((scenes, sizes, off, pts, masks, o))
    ctx.set_boxes([sc["corners_velo"] for sc in runs[0][0]])       # same boxes for every run (tables are per context)
    torch.cuda.synchronize(dev)
    for scenes, sizes, off, pts, masks, o in runs:
        ctx.set_masks(masks, lend=lent)
        ctx.run_device(pts, off, inst_cap=int(off[-1]), **o)
    ctx.sync()
    torch.cuda.synchronize(dev)
    for scenes, sizes, off, pts, masks, o in runs:
        sm = np.frombuffer(o["summary"].cpu().numpy().tobytes(), SUMMARY_DTYPE)
        uv, lab = o["uv"].cpu().numpy(), o["label_bits"].cpu().numpy().view(np.uint32)
        vidx, iidx, cmb = o["valid_idx"].cpu().numpy(), o["inst_idx"].cpu().numpy(), o["count_mb"].cpu().numpy()
        for f, (sc, nf) in enumerate(zip(scenes, sizes)):
            a, b = int(off[f]), int(off[f + 1])
            limg = orc.pack_masks(sc["masks"], 0, H, W)
            ref = orc.run(sc["points"][:nf], T, K, W, H, 0.0, 30.0, label_img=limg, M=M, corners=runs[0][0][f]["corners_velo"],
                          want_float=False)
            assert np.array_equal(uv[a:b, 0], ref["u"]) and np.array_equal(uv[a:b, 1], ref["v"])
            assert np.array_equal(lab[a:b], ref["label_bits"])
            assert int(sm[f]["n_valid"]) == ref["n_valid"]
            assert np.array_equal(vidx[a:a + ref["n_valid"]], ref["valid_idx"])
            assert np.array_equal(sm[f]["inst_count"][:M], ref["inst_count"])
            for m in range(M):
                lo, hi = int(sm[f]["inst_off"][m]), int(sm[f]["inst_off"][m + 1])
                assert np.array_equal(iidx[f, lo:hi], ref["inst_lists"][m])
            assert np.array_equal(cmb[M * Bx * f:M * Bx * (f + 1)].reshape(M, Bx), ref["count_mb"])
            assert np.array_equal(sm[f]["best_box"][:M], ref["best_box"]) and np.array_equal(sm[f]["best_cnt"][:M], ref["best_cnt"])
    ctx.close()


@pytest.mark.parametrize("n, M, Bx", [(150_000, 6, 9), (1_000_000, 8, 32)], ids=["150k", "configs4_1M_8masks_32boxes"])
def test_hipgraph_replay_matches_oracle(calib, n, M, Bx):
    """BASELINE configs[4]: the per-frame launch set (mask pack + 1 erosion, project+label, lists + box counts,
    finalize) captured once into a hipGraph and replayed on new frame contents in the same buffers; every replay
    must equal the oracle.  The second case is configs[4]'s own size: 1 M points, 8 eroded masks, 32 boxes."""
    import torch
    from lidar_object_detection_amd import synthetic as S
    from lidar_object_detection_amd._native import LpfContext, SUMMARY_DTYPE
    _, T, K, W, H = S.default_calibration(calib)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(dev)
    with torch.cuda.stream(stream):
        ctx = LpfContext(0)
        ctx.set_stream(stream.cuda_stream)
        ctx.set_camera(T, K, W, H, 0.0, 50.0)
        sc0 = S.scene(n, n_masks=M, n_boxes=Bx, seed=500)
        ctx.set_boxes(sc0["corners_velo"])
        pts = torch.from_numpy(sc0["points"]).to(dev)
        masks = torch.from_numpy(sc0["masks"]).to(dev)
        o = dict(uv=torch.empty((n, 2), dtype=torch.int32, device=dev), label_bits=torch.empty(n, dtype=torch.int32, device=dev),
                 valid_idx=torch.empty(n, dtype=torch.int64, device=dev), inst_idx=torch.empty(n, dtype=torch.int64, device=dev),
                 count_mb=torch.zeros(M * Bx, dtype=torch.int32, device=dev),
                 summary=torch.zeros(SUMMARY_DTYPE.itemsize, dtype=torch.uint8, device=dev))
        off = np.array([0, n], np.int64)
        step = ctx.make_device_step(pts, off, masks_u8=masks.unsqueeze(0), erode_iters=1, inst_cap=n, **o)
        step()                                              # warm: allocations + table uploads happen here
        ctx.sync()
        ctx.graph_begin()
        step()
        g = ctx.graph_end()
        for k in range(3):
            sc = S.scene(n, n_masks=M, n_boxes=Bx, seed=501 + k)
            pts.copy_(torch.from_numpy(sc["points"]))
            masks.copy_(torch.from_numpy(sc["masks"]))
            stream.synchronize()
            ctx.graph_launch(g)
            ctx.sync()
            sm = np.frombuffer(o["summary"].cpu().numpy().tobytes(), SUMMARY_DTYPE)[0]
            limg = orc.pack_masks(sc["masks"], 1, H, W)
            ref = orc.run(sc["points"], T, K, W, H, 0.0, 50.0, label_img=limg, M=M, corners=sc0["corners_velo"], want_float=False)
            uv = o["uv"].cpu().numpy()
            assert np.array_equal(uv[:, 0], ref["u"]) and np.array_equal(uv[:, 1], ref["v"])
            assert np.array_equal(o["label_bits"].cpu().numpy().view(np.uint32), ref["label_bits"])
            assert int(sm["n_valid"]) == ref["n_valid"]
            assert np.array_equal(o["valid_idx"].cpu().numpy()[:ref["n_valid"]], ref["valid_idx"])
            assert np.array_equal(sm["inst_count"][:M], ref["inst_count"])
            assert np.array_equal(o["count_mb"].cpu().numpy().reshape(M, Bx), ref["count_mb"])
            assert np.array_equal(sm["best_box"][:M], ref["best_box"])
        ctx.graph_destroy(g)
        ctx.close()


@pytest.mark.parametrize("mode", ["u8", "f32_v3"])
@pytest.mark.parametrize("iters", [0, 1, 2])
def test_odd_image_size_takes_the_lds_tile_pack(ctx, mode, iters):
    """W*H not a multiple of 16 -> the general LDS-tile pack (fused first erosion) instead of the
    16-pixel streaming pack; the projection is then checked on that image too."""
    rng = np.random.default_rng(17)
    W, H, M = 77, 45, 11                                  # 11 masks -> uint16 label image
    base = (rng.random((M, H, W)) < 0.85).astype(np.uint8)
    base[0] = 1
    T = np.eye(4)
    K = np.array([[30.0, 0, W / 2], [0, 30.0, H / 2], [0, 0, 1.0]])
    ctx.set_camera(T, K, W, H, 0.0, 50.0)
    if mode == "u8":
        ctx.set_masks(base, erode_iters=iters)
        member = base
    else:
        m = base.astype(np.float32)
        ctx.set_masks(m, erode_iters=iters, v3_pipeline=True)
        member = orc.binarize_f32(m, 1)
    want = orc.pack_masks(member, iters, H, W)
    assert np.array_equal(ctx.get_label_image()[0], want)
    pts = np.zeros((20000, 4), np.float32)
    pts[:, 0] = rng.uniform(-4, 4, len(pts)); pts[:, 1] = rng.uniform(-3, 3, len(pts)); pts[:, 2] = rng.uniform(0.5, 6, len(pts))
    ctx.clear_boxes()
    r = ctx.run(pts, want_float=False)
    o = orc.run(pts, T, K, W, H, 0.0, 50.0, label_img=want, M=M, want_float=False)
    _compare(r, o, M, want_float=False)


def test_frame_with_more_than_1024_segments(ctx, calib):
    """4.6 M points in ONE frame = 1123 segments: the general (two-phase) segment scan instead of the
    register-resident one; masks and boxes on, every list compared."""
    from lidar_object_detection_amd import synthetic as S
    _, T, K, W, H = S.default_calibration(calib)
    sc = S.scene(4_600_000, n_masks=4, n_boxes=6, seed=77)
    ctx.set_camera(T, K, W, H, 0.0, 50.0)
    ctx.set_masks(sc["masks"])
    ctx.set_boxes(sc["corners_velo"])
    r = ctx.run(sc["points"], want_float=False)
    lab = orc.pack_masks(sc["masks"], 0, H, W)
    o = orc.run(sc["points"], T, K, W, H, 0.0, 50.0, label_img=lab, M=4, corners=sc["corners_velo"], want_float=False)
    _compare(r, o, 4, want_float=False)


def test_error_codes_are_loud(calib):
    """No silent fallbacks: wrong call order / shapes give LpfError with the library's message."""
    from lidar_object_detection_amd._native import LpfContext, LpfError
    from lidar_object_detection_amd import synthetic as S
    _, T, K, W, H = S.default_calibration(calib)
    c = LpfContext(0)
    pts = S.synthetic_cloud(1000, seed=1)
    with pytest.raises(LpfError, match="set_camera"):
        c.run(pts)
    with pytest.raises(LpfError, match="set_camera"):
        c.set_masks(np.zeros((1, 8, 8), np.uint8)) if False else c._check(c._lib.lpf_set_masks_u8(c._h, None, 1, 1, 0, 0))
    c.set_camera(T, K, W, H, 0.0, 50.0)
    with pytest.raises(ValueError):
        c.set_masks(np.zeros((2, H + 1, W), np.uint8))                      # wrong image size (wrapper check)
    with pytest.raises(LpfError, match="M=33"):
        c._check(c._lib.lpf_set_masks_u8(c._h, np.zeros((33, H, W), np.uint8).ctypes.data, 1, 33, 0, 0))
    c.set_masks(np.zeros((2, 3, H, W), np.uint8))                           # masks for 2 frames ...
    with pytest.raises(LpfError, match="masks were set for 2 frames"):
        c.run(pts)                                                          # ... but a 1-frame run
    c.clear_masks()
    c.set_boxes([np.zeros((1, 8, 3)), np.zeros((2, 8, 3))])
    with pytest.raises(LpfError, match="boxes were set for 2 frames"):
        c.run(pts)
    c.clear_boxes()
    r = c.run(pts)                                                           # and it still works afterwards
    assert r["n_valid"] == orc.run(pts, T, K, W, H, 0.0, 50.0, want_float=False)["n_valid"]
    with pytest.raises(LpfError, match="need valid_idx"):
        from lidar_object_detection_amd._native import Outputs
        import ctypes
        o2 = Outputs(); buf = np.empty((1000, 2), np.int32); o2.uv_valid = buf.ctypes.data
        c._check(c._lib.lpf_run(c._h, pts.ctypes.data, 1000, 0, ctypes.byref(o2)))
    with pytest.raises(LpfError, match="inst_cap"):
        from lidar_object_detection_amd._native import Outputs
        import ctypes
        o = Outputs(); o.inst_idx = 1; o.inst_cap = 0
        c._check(c._lib.lpf_run(c._h, pts.ctypes.data, 1000, 0, ctypes.byref(o)))
    c.close()


def test_degenerate_boxes_and_nan_points(ctx, calib):
    """Zero-size and NaN boxes (vv == 0 / NaN: the reference's division gives nan/inf -> outside), NaN / inf
    points: exact agreement with the oracle, no fault."""
    from lidar_object_detection_amd import synthetic as S
    _, T, K, W, H = S.default_calibration(calib)
    sc = S.scene(60000, n_masks=3, n_boxes=5, seed=88)
    corners = sc["corners_velo"].copy()
    corners[1] = corners[1][0]                           # all 8 corners identical: vv = 0 on every axis
    corners[2][3] = corners[2][0]                        # one degenerate axis
    corners[3][4, 0] = np.nan
    pts = sc["points"].copy()
    pts[::97, 0] = np.nan
    pts[5::101, 2] = np.inf
    pts[7::103, 1] = -np.inf
    masks = sc["masks"].copy()
    masks[0] = 1
    ctx.set_camera(T, K, W, H, 0.0, 50.0)
    ctx.set_masks(masks)
    ctx.set_boxes(corners)
    r = ctx.run(pts, want_float=True)
    lab = orc.pack_masks(masks, 0, H, W)
    o = orc.run(pts, T, K, W, H, 0.0, 50.0, label_img=lab, M=3, corners=corners)
    _compare(r, o, 3)


FULL = [r["frame"] for r in golden_frames().get("full_frames", [])]


@pytest.mark.parametrize("frame", FULL)
@pytest.mark.parametrize("tag", ["rect5_d50", "rect5_d30", "edge_d50"])
def test_full_size_golden_frames(ctx, calib, frame, tag):
    """Sample frames 1461, 2098 and 2449 at FULL size against the reference-generated digests: real scan order (segments
    on cars hold hundreds of masked points), up to 133 visible boxes (two candidate words per cell), and with the nine
    'edge' masks M x B = 1197 > 1024 (the per-hit global-atomic path of the box count)."""
    g = load_golden_full(frame)
    kind, dmax = tag.split("_d")
    W, H = int(calib["width"]), int(calib["height"])
    masks = unpack_masks(g, kind, H, W)
    M = masks.shape[0]
    ctx.set_camera(calib["TrVeloToRect"], calib["K"], W, H, 0.0, float(dmax))
    ctx.set_masks(masks)
    ctx.set_boxes(g["corners_velo"])
    r = ctx.run(g["points"], want_float=True, want_valid_uv=True)
    check_full(g, "u", r["u"], np.int64)
    check_full(g, "v", r["v"], np.int64)
    check_full(g, "valid_idx_d" + dmax, r["valid_idx"], np.int64)
    check_full(g, "inst_cat_" + tag, np.concatenate(r["inst_lists"]), np.int64)
    check_full(g, "depth_s", r["depth"][::FS], np.float64)
    check_full(g, "uf_s", r["uf"][::FS], np.float64)
    check_full(g, "vf_s", r["vf"][::FS], np.float64)
    check_full(g, "bg_assigned_" + tag, np.packbits(r["label_valid"] != 0), np.uint8)
    assert np.array_equal(r["inst_count"], g["inst_count_" + tag])
    assert np.array_equal(r["count_mb"], g["count_mb_" + tag])
    rows = [m for m in range(M) if r["inst_count"][m] > 0]
    assert np.array_equal(np.array([r["best_box"][m] if r["best_cnt"][m] >= 10 else -1 for m in rows], np.int64),
                          g["stats_matched_bbox_id_" + tag])
    assert np.array_equal(np.array([r["best_cnt"][m] if r["best_cnt"][m] >= 10 else 0 for m in rows], np.int64),
                          g["stats_points_inside_bbox_" + tag])
    ctx.set_boxes(g["corners_velo"], oriented=False)
    assert np.array_equal(ctx.run(g["points"])["count_mb"], g["count_mb_aabb_" + tag])
    # every annotated box of the frame (up to 314: five candidate words per cell), prepared on the device
    (vis, cv, _, _), = ctx.set_boxes_cam0(g["corners_cam0_raw"], np.linalg.inv(calib["TrVeloToCam"]))
    assert np.array_equal(np.flatnonzero(vis), g["visible_pos"]) and np.array_equal(cv[vis], g["corners_velo"])
    cm = ctx.run(g["points"])["count_mb"]
    assert np.array_equal(cm[:, vis], g["count_mb_" + tag]) and not cm[:, ~vis].any()
