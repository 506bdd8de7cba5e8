"""The CPU oracle (oracle/lpf_oracle.c) against golden vectors produced by the
reference's own functions (tests/golden/make_golden.py).  Bit-exact on every integer
output; pre-rounding (u,v) floats within 1e-5 (they are in fact bit-equal here)."""
import numpy as np
import pytest

from conftest import check_full, golden_frames, load_golden, load_golden_full, unpack_masks
from oracle import cpu_oracle as orc

FRAMES = [r for r in golden_frames()["frames"]]
FS = golden_frames()["float_stride"]
I32 = np.iinfo(np.int32)


def _sat32(a):
    return np.clip(a, I32.min, I32.max).astype(np.int32)


@pytest.mark.parametrize("rec", FRAMES, ids=lambda r: "f%d" % r["frame"])
def test_projection_matches_reference(rec, calib):
    g = load_golden(rec["frame"])
    if "u" not in g:
        pytest.skip("frame has no boxes: reference skips it before projecting (V3:557-558)")
    o = orc.project(g["points"], calib["TrVeloToRect"], calib["K"][:, :3])
    assert np.array_equal(o["u64"], g["u"])
    assert np.array_equal(o["v64"], g["v"])
    assert np.array_equal(o["u32"], _sat32(g["u"]))
    assert np.array_equal(o["v32"], _sat32(g["v"]))
    # tolerance stated by BASELINE.json north_star: 1e-5 on projected (u,v) floats
    for k in ("depth", "uf", "vf"):
        a, b = o[k][::FS], g[k + "_s"]
        fin = np.isfinite(b)
        assert np.array_equal(np.isfinite(a), fin)
        assert np.max(np.abs(a[fin] - b[fin]) / np.maximum(1.0, np.abs(b[fin]))) <= 1e-5
        assert np.array_equal(a, b), "expected bit-equality with NumPy/OpenBLAS on this host"


@pytest.mark.parametrize("rec", FRAMES, ids=lambda r: "f%d" % r["frame"])
@pytest.mark.parametrize("tag", ["rect5_d50", "rect5_d30", "edge_d50"])
def test_full_path_matches_reference(rec, tag, calib):
    g = load_golden(rec["frame"])
    if "u" not in g:
        pytest.skip("frame skipped by the reference")
    kind, dmax = tag.split("_d")
    W, H = int(calib["width"]), int(calib["height"])
    masks = unpack_masks(g, kind, H, W)
    M = masks.shape[0]
    lab = orc.pack_masks(orc.binarize_f32(masks, 0), 0, H, W)
    o = orc.run(g["points"], calib["TrVeloToRect"], calib["K"][:, :3], W, H, 0.0, float(dmax),
                label_img=lab, M=M, corners=g["corners_velo"], oriented=True)
    assert np.array_equal(o["valid_idx"], g["valid_idx_d" + dmax])
    assert np.array_equal(o["inst_count"], g["inst_count_" + tag])
    cat = np.concatenate(o["inst_lists"]) if M else np.zeros(0, np.int64)
    assert np.array_equal(cat, g["inst_cat_" + tag])
    assert np.array_equal(o["count_mb"], g["count_mb_" + tag])
    # V4's bg_assigned == (label != 0) on the valid points
    bg = np.unpackbits(g["bg_assigned_" + tag])[:o["n_valid"]].astype(bool)
    assert np.array_equal(o["label_bits"][o["valid_idx"]] != 0, bg)
    # best-box scan -> the reference's stats rows
    tot = o["inst_count"]
    rows = [m for m in range(M) if tot[m] > 0] if g["corners_velo"].shape[0] else []
    assert np.array_equal(np.array(rows, np.int64), g["stats_car_id_" + tag])
    matched = np.array([o["best_box"][m] if o["best_cnt"][m] >= 10 else -1 for m in rows], np.int64)
    inside = np.array([o["best_cnt"][m] if o["best_cnt"][m] >= 10 else 0 for m in rows], np.int64)
    assert np.array_equal(matched, g["stats_matched_bbox_id_" + tag])
    assert np.array_equal(inside, g["stats_points_inside_bbox_" + tag])
    assert np.array_equal(tot[rows], g["stats_total_points_" + tag])
    # AABB variant (use_oriented=False)
    oa = orc.run(g["points"], calib["TrVeloToRect"], calib["K"][:, :3], W, H, 0.0, float(dmax),
                 label_img=lab, M=M, corners=g["corners_velo"], oriented=False, want_float=False)
    assert np.array_equal(oa["count_mb"], g["count_mb_aabb_" + tag])


FULL = [r["frame"] for r in golden_frames().get("full_frames", [])]


@pytest.mark.parametrize("frame", FULL)
@pytest.mark.parametrize("tag", ["rect5_d50", "rect5_d30", "edge_d50"])
def test_full_size_frames_match_reference(frame, tag, calib):
    """Frames 1461, 2098, 2449 at FULL size (real scan order: dense segments; up to 133 visible boxes): the reference's
    per-point outputs are committed as SHA-256 digests, the small ones as they are."""
    g = load_golden_full(frame)
    kind, dmax = tag.split("_d")
    W, H = int(calib["width"]), int(calib["height"])
    masks = unpack_masks(g, kind, H, W)
    M = masks.shape[0]
    lab = orc.pack_masks(orc.binarize_f32(masks, 0), 0, H, W)
    o = orc.run(g["points"], calib["TrVeloToRect"], calib["K"][:, :3], W, H, 0.0, float(dmax),
                label_img=lab, M=M, corners=g["corners_velo"], oriented=True)
    check_full(g, "u", o["u64"] if "u64" in o else o["u"], np.int64)
    check_full(g, "v", o["v64"] if "v64" in o else o["v"], np.int64)
    check_full(g, "valid_idx_d" + dmax, o["valid_idx"], np.int64)
    check_full(g, "inst_cat_" + tag, np.concatenate(o["inst_lists"]), np.int64)
    check_full(g, "depth_s", o["depth"][::FS], np.float64)
    check_full(g, "uf_s", o["uf"][::FS], np.float64)
    assert np.array_equal(o["inst_count"], g["inst_count_" + tag])
    assert np.array_equal(o["count_mb"], g["count_mb_" + tag])
    check_full(g, "bg_assigned_" + tag, np.packbits(o["label_bits"][o["valid_idx"]] != 0), np.uint8)
    rows = [m for m in range(M) if o["inst_count"][m] > 0]
    assert np.array_equal(np.array([o["best_box"][m] if o["best_cnt"][m] >= 10 else -1 for m in rows], np.int64),
                          g["stats_matched_bbox_id_" + tag])
    oa = orc.run(g["points"], calib["TrVeloToRect"], calib["K"][:, :3], W, H, 0.0, float(dmax),
                 label_img=lab, M=M, corners=g["corners_velo"], oriented=False, want_float=False)
    assert np.array_equal(oa["count_mb"], g["count_mb_aabb_" + tag])


def test_frame100_is_the_survey_frame():
    rec = [r for r in FRAMES if r["frame"] == 100][0]
    assert rec["n_points"] == 109355 and rec["n_valid_d50"] == 25662 and rec["n_valid_d30"] == 23293
    assert rec["n_boxes_raw"] == 31 and rec["n_boxes_visible"] == 25 and rec["n_masks_rect5"] == 5


def test_frame_2717_has_no_boxes():
    rec = [r for r in FRAMES if r["frame"] == 2717][0]
    assert rec.get("skipped") == "no boxes"


def test_depth_maps_match_reference_loop(calib):
    """seg_with_pointcloud.py:145-170 (per-car depth maps, last writer wins) through the oracle's single
    last-writer image: depthMap_i == where(mask_i > 0.5, D, 0)."""
    g = load_golden(100)
    W, H = int(calib["width"]), int(calib["height"])
    D, win = orc.depth_image(g["points"], calib["TrVeloToRect"], calib["K"][:, :3], W, H, 0.0, 30.0)
    masks = unpack_masks(g, "rect5", H, W)
    off = g["depthmap_off_rect5"]
    for i, m in enumerate(masks):
        dm = np.where(m > 0.5, D, 0.0)
        flat = np.flatnonzero(dm)
        assert np.array_equal(flat, g["depthmap_idx_rect5"][off[i]:off[i + 1]])
        assert np.array_equal(dm.ravel()[flat], g["depthmap_val_rect5"][off[i]:off[i + 1]])
    vi = g["valid_idx_d30"]
    assert set(np.unique(win[win >= 0])) <= set(vi.tolist()) and (win >= 0).sum() == np.count_nonzero(D)


@pytest.mark.parametrize("rec", FRAMES, ids=lambda r: "f%d" % r["frame"])
def test_box_preparation_restatement_matches_reference(rec, calib):
    """oracle/numpy_path.prepare_boxes (filter_visible_bboxes + transform_bboxes_to_velodyne, V3:121-140 / V3:41-52) against
    what the reference's own functions produced for every sample frame: which boxes are kept, and their velodyne corners."""
    from oracle import numpy_path as npp
    g = load_golden(rec["frame"])
    if "corners_cam0_raw" not in g:
        pytest.skip("no boxes for this frame")
    vis, velo = npp.prepare_boxes(g["corners_cam0_raw"], np.asarray(calib["K"])[:, :3], int(calib["width"]), int(calib["height"]),
                                  calib["TrVeloToCam"])
    assert np.array_equal(np.flatnonzero(vis), g["visible_pos"])
    assert np.array_equal(velo[g["visible_pos"]], g["corners_velo"])
