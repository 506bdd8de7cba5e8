"""The reference-shaped Python API (lidar_object_detection_amd.pipeline) on the GPU against the
golden vectors produced by the reference's own functions."""
import contextlib
import io
import json
import os

import numpy as np
import pytest

from conftest import golden_frames, load_golden, unpack_masks
from lidar_object_detection_amd import kitti360, pipeline
from lidar_object_detection_amd._native import LpfContext, SUMMARY_DTYPE
from oracle import cpu_oracle as orc

pytestmark = pytest.mark.gpu
FRAMES = [r for r in golden_frames()["frames"] if "skipped" not in r]
I32 = np.iinfo(np.int32)


def _camera(calib):
    return kitti360.CameraPerspective.from_arrays(calib["K"], calib["R_rect"], int(calib["width"]), int(calib["height"]))


def _boxes(g):
    return [{"corners_cam0": None, "corners_velo": c.tolist()} for c in g["corners_velo"]]


def _quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


@pytest.mark.parametrize("rec", FRAMES[:4], ids=lambda r: "f%d" % r["frame"])
def test_project_points(rec, calib):
    g = load_golden(rec["frame"])
    cam = _camera(calib)
    for dmax in (50, 30):
        u, v, depth, vi = pipeline.project_points(g["points"], calib["TrVeloToRect"], cam, depth_max=dmax)
        assert u.dtype == np.int64 and v.dtype == np.int64 and vi.dtype == np.int64
        assert np.array_equal(u, np.clip(g["u"], I32.min, I32.max)) and np.array_equal(v, np.clip(g["v"], I32.min, I32.max))
        assert np.array_equal(vi, g["valid_idx_d%d" % dmax])


@pytest.mark.parametrize("rec", FRAMES, ids=lambda r: "f%d" % r["frame"])
@pytest.mark.parametrize("tag", ["rect5_d50", "edge_d50"])
def test_function_level_api_matches_reference(rec, tag, calib):
    g = load_golden(rec["frame"])
    cam = _camera(calib)
    kind, dmax = tag.split("_d")
    masks = unpack_masks(g, kind, cam.height, cam.width)
    vi = g["valid_idx_d" + dmax]
    pv, uv, vv = g["points"][vi, :3], g["u"][vi], g["v"][vi]
    sets = pipeline.extract_car_points_by_mask(pv, uv, vv, list(masks), cam)
    counts = g["inst_count_" + tag]
    assert [len(s) for s in sets] == counts.tolist()
    off = np.concatenate([[0], np.cumsum(counts)])
    for m, s in enumerate(sets):
        want = g["points"][g["inst_cat_" + tag][off[m]:off[m + 1]], :3]
        assert s.shape == (counts[m], 3) and np.array_equal(s, want)
        assert s.dtype == (np.float32 if counts[m] else np.float64)          # (0,3) float64 for empty masks, V3:231
    boxes = _boxes(g)
    colors = pipeline.default_colors(len(sets))
    for style in ("v3", "cvs"):
        st = _quiet(pipeline.calculate_car_point_statistics, sets, boxes, colors, 10, True, style)
        for key in ("car_id", "matched_bbox_id", "total_points", "points_inside_bbox", "points_outside_bbox"):
            assert [r[key] for r in st] == g["stats_%s_%s" % (key, tag)].tolist(), (style, key)
        assert np.array_equal(np.array([r["inside_percentage"] for r in st]), g["stats_inside_percentage_" + tag])
        if style == "v3":
            for r in st:
                if r["matched_bbox_id"] >= 0:
                    want = orc.points_in_box(r["car_points"], g["corners_velo"][r["matched_bbox_id"]], True)
                    assert np.array_equal(r["inside_mask"], want) and int(want.sum()) == r["points_inside_bbox"]
                    assert np.array_equal(r["corners_velo"], g["corners_velo"][r["matched_bbox_id"]])
                else:
                    assert r["inside_mask"] is None and r["corners_velo"] is None
        else:
            assert all("car_points" not in r for r in st)
    st = _quiet(pipeline.calculate_car_point_statistics, sets, boxes, colors, 10, False, "cvs")
    assert [r["matched_bbox_id"] for r in st] == g["stats_aabb_matched_bbox_id_" + tag].tolist()
    assert [r["points_inside_bbox"] for r in st] == g["stats_aabb_points_inside_bbox_" + tag].tolist()
    if tag == "rect5_d50":
        mp = _quiet(pipeline.match_car_points_to_bboxes, sets, boxes, colors, 10, True)
        assert [t[2] for t in mp] == g["matchpairs_count_" + tag].tolist()
        assert np.array_equal(np.array([t[0] for t in mp]).reshape(-1, 8, 3), g["matchpairs_corners_" + tag])


def test_point_in_box_operators(calib):
    g = load_golden(100)
    pts = g["points"][::7, :3]
    for b in (0, 3, 11):
        c = g["corners_velo"][b]
        assert np.array_equal(pipeline.oriented_point_in_bbox(pts, c), orc.points_in_box(pts, c, True))
        assert np.array_equal(pipeline.point_in_bbox(pts, c), orc.points_in_box(pts, c, False))
    e = pipeline.oriented_point_in_bbox(np.zeros((0, 3), np.float32), g["corners_velo"][0])
    assert e.shape == (0,) and e.dtype == np.float64                     # np.array([]) as the reference, V3:179-180
    with pytest.raises(TypeError):
        pipeline.oriented_point_in_bbox(np.array([[0.1, 0.2, 0.3]]), g["corners_velo"][0])   # not float32-representable


def test_slab_quotient_that_underflows_to_minus_zero(calib):
    """V3:195-202 forms t = d / |v|^2 and accepts 0 <= t <= 1.  With a huge |v|^2 a tiny negative d gives t = -0.0, which
    the reference counts as inside; a division-free test (d >= 0) would say outside.  The kernel only skips the division
    where the two provably agree (1e-100 <= |v|^2 <= 1e100, |d| >= 1e-200 or d == 0) -- this box lies outside that range."""
    def box(c0, v1, v2, v3):                                 # corners such that c1-c0, c3-c0, c4-c0 are the three slab vectors
        c = np.zeros((8, 3))
        c[0] = c0
        c[1], c[3], c[4] = c0 + v1, c0 + v2, c0 + v3
        return c
    pts = np.array([[0.0, 0.5, 0.5], [0.0, 0.5, 1.5], [0.0, -0.5, 0.5]], np.float32)
    for vx in (1e60, 1e45, 1e30):                            # |v|^2 = 1e120 (quotient path), 1e90 and 1e60 (division-free path)
        c = box(np.array([1e-305, 0.0, 0.0]), np.array([vx, 0.0, 0.0]), np.array([0.0, 1.0, 0.0]), np.array([0.0, 0.0, 1.0]))
        want = orc.points_in_box(pts, c, True)
        t = (pts[:, 0].astype(np.float64) - c[0, 0]) * vx / (vx * vx)
        if vx == 1e60:
            assert t[0] == 0.0 and np.signbit(t[0]) and want.tolist() == [True, False, False]     # the counter-example is one
        assert np.array_equal(pipeline.oriented_point_in_bbox(pts, c), want), vx


def test_all_sample_frames_in_one_batch(calib):
    """BASELINE configs[3] on one GPU: every sample frame (ragged N, M, B) in ONE batched call."""
    cam = _camera(calib)
    items, gold = [], []
    for rec in FRAMES:
        g = load_golden(rec["frame"])
        masks = unpack_masks(g, "rect5", cam.height, cam.width)
        items.append(pipeline.FrameInputs(rec["frame"], g["points"], masks, _boxes(g)))
        gold.append(g)
    res = pipeline.run_frames(items, calib["TrVeloToRect"], cam, depth_max=50.0)
    rows = 0
    for r, g in zip(res, gold):
        tag = "rect5_d50"
        assert np.array_equal(r["valid_indices"], g["valid_idx_d50"])
        assert np.array_equal(r["count_mb"], g["count_mb_" + tag])
        assert [len(s) for s in r["car_point_sets"]] == g["inst_count_" + tag].tolist()
        for key in ("car_id", "matched_bbox_id", "total_points", "points_inside_bbox", "points_outside_bbox"):
            assert [d[key] for d in r["car_statistics"]] == g["stats_%s_%s" % (key, tag)].tolist(), (r["frame"], key)
        bg = np.unpackbits(g["bg_assigned_" + tag])[:len(r["valid_indices"])].astype(bool)
        assert np.array_equal(r["bg_assigned"], bg)
        rows += len(r["car_statistics"])
    assert rows > 20


def test_process_frames_entry_point(calib, tmp_path, monkeypatch):
    """cvs_erosion.process_frames end to end on a dataset tree rebuilt from the fixtures: skip rules,
    CSV rows in frame order, aggregate printout."""
    cam = _camera(calib)
    root = tmp_path / "KITTI360_sample"
    seq = "2013_05_28_drive_0000_sync"
    (root / "data_3d_raw" / seq / "velodyne_points" / "data").mkdir(parents=True)
    (root / "bboxes_3D_cam0").mkdir()
    (root / "data_2d_raw" / seq / "image_00" / "data_rect").mkdir(parents=True)
    recs = golden_frames()["frames"]
    use = [r for r in recs if r["frame"] in (100, 250, 570, 2717, 2939)]
    masks_of, expect = {}, []
    for r in use:
        g = load_golden(r["frame"])
        g["points"].tofile(str(root / "data_3d_raw" / seq / "velodyne_points" / "data" / ("%010d.bin" % r["frame"])))
        (root / "data_2d_raw" / seq / "image_00" / "data_rect" / ("%010d.png" % r["frame"])).write_bytes(b"")
        if "skipped" in r:
            continue                                                     # 2717: no BBoxes json -> frame dropped
        raw = [{"index": int(i), "corners_cam0": c.tolist()} for i, c in zip(g["box_index_raw"], g["corners_cam0_raw"])]
        if r["frame"] == 250:                                            # a key beyond the plain schema: the library's parser leaves this
            for b in raw:                                                # file to json.load (lpf_parse_boxes_json: OTHER), same rows
                b["label"] = "car"
        (root / "bboxes_3D_cam0" / ("BBoxes_%d.json" % r["frame"])).write_text(json.dumps(raw, indent=2 if r["frame"] == 570 else None))
        masks_of[r["frame"]] = unpack_masks(g, "rect5", cam.height, cam.width)
        if r["frame"] == 570:                                            # ... and a frame whose box file is an empty list: skipped (cvs:334-335)
            g["points"].tofile(str(root / "data_3d_raw" / seq / "velodyne_points" / "data" / ("%010d.bin" % 571)))
            (root / "data_2d_raw" / seq / "image_00" / "data_rect" / ("%010d.png" % 571)).write_bytes(b"")
            (root / "bboxes_3D_cam0" / "BBoxes_571.json").write_text("[]")
            masks_of[571] = masks_of[570]
        for i in range(len(g["stats_car_id_rect5_d50"])):
            expect.append((r["frame"], int(g["stats_car_id_rect5_d50"][i]), int(g["stats_matched_bbox_id_rect5_d50"][i]),
                           int(g["stats_total_points_rect5_d50"][i]), int(g["stats_points_inside_bbox_rect5_d50"][i])))
    velo = kitti360.Kitti360Viewer3DRaw(seq=0, root_dir=str(root))
    monkeypatch.setattr(pipeline, "sequence_setup",
                        lambda path, s=0, c=0: (seq, cam, calib["TrVeloToCam"], calib["TrVeloToRect"], velo))

    def segmenter(image_path):
        frame = int(os.path.basename(image_path).split(".")[0])
        m = masks_of[frame]
        return None, m, pipeline.default_colors(len(m)), np.zeros((len(m), 4), np.float32), np.ones(len(m))

    csv_path = str(tmp_path / "results" / "master_car_statistics.csv")
    with contextlib.redirect_stdout(io.StringIO()) as out:
        df = pipeline.process_frames(0, 0, segmenter=segmenter, image_loader=lambda p: p, kitti360_path=str(root),
                                     master_csv_path=csv_path, timestamp="T", read_ahead=False, batch_frames=2)   # (three batches)
    got = [(int(a), int(b), int(c), int(d), int(e)) for a, b, c, d, e in
           zip(df["frame"], df["car_id"], df["matched_bbox_id"], df["total_points"], df["points_inside_bbox"])]
    assert got == expect and len(expect) > 5
    text = out.getvalue()
    assert "Found 6 frames to process" in text and "No bounding boxes found" in text and "OVERALL ANALYSIS" in text
    # the same run frame by frame with the native read-ahead reader (the default): same rows, same file
    csv2 = str(tmp_path / "results2" / "master_car_statistics.csv")
    with contextlib.redirect_stdout(io.StringIO()) as out2:
        df2 = pipeline.process_frames(0, 0, segmenter=segmenter, image_loader=lambda p: p, kitti360_path=str(root),
                                      master_csv_path=csv2, timestamp="T")
    assert open(csv2).read() == open(csv_path).read() and df2.equals(df)
    assert "Found 6 frames to process" in out2.getvalue() and "OVERALL ANALYSIS" in out2.getvalue()
    assert "No bounding boxes found" in out2.getvalue() and sorted(out2.getvalue().replace("results2", "results").splitlines()) == sorted(text.splitlines())   # the same lines


def test_boxes_parsed_by_the_library_equal_the_json_path(calib, tmp_path):
    """prepare_boxes_from_arrays (box file parsed by lpf_parse_boxes_json / the reader's worker) against prepare_boxes on json.load's
    dicts: the same visible boxes in the same order, the same corners bit for bit, the same 2D boxes."""
    from lidar_object_detection_amd import _native
    cam = _camera(calib)
    for frame in (100, 2449):
        g = load_golden(frame)
        raw = [{"index": int(i), "corners_cam0": c.tolist()} for i, c in zip(g["box_index_raw"], g["corners_cam0_raw"])]
        p = tmp_path / ("BBoxes_%d.json" % frame)
        p.write_text(json.dumps(raw))
        st, idx, cs = _native.parse_boxes_file(str(p))
        assert st == _native.BOXES_PARSED
        for keep_all in (False, True):
            a = pipeline.prepare_boxes_from_arrays(idx, cs, cam, calib["TrVeloToCam"], keep_all=keep_all)
            b = pipeline.prepare_boxes(json.load(open(p)), cam, calib["TrVeloToCam"], keep_all=keep_all)
            assert len(a) == len(b) > 0 and [d["index"] for d in a] == [d["index"] for d in b]
            assert np.array_equal(np.array([d["corners_velo"] for d in a]), np.array([d["corners_velo"] for d in b]))
            assert [d["_bbox2d"] for d in a] == [d["_bbox2d"] for d in b] and [d["_front"] for d in a] == [d["_front"] for d in b]
            ca, pa = pipeline._corners_velo(a)
            cb, pb = pipeline._corners_velo(b)
            assert np.array_equal(ca, cb) and pa == pb == list(range(len(a)))
            assert np.array_equal(pipeline._corners_velo(list(b))[0], cb)            # (a plain list of the same dicts: the walk gives the same)
    assert pipeline._boxes_of_file(str(tmp_path / "BBoxes_none.json"), cam, calib["TrVeloToCam"]) is None


@pytest.mark.parametrize("rec", FRAMES, ids=lambda r: "f%d" % r["frame"])
def test_prepare_boxes_on_gpu(rec, calib):
    """visibility filter + cam->velo transform + projected 2D boxes on the GPU, bit-exact against
    the reference's filter_visible_bboxes / transform_bboxes_to_velodyne / match_detections_to_bboxes outputs."""
    g = load_golden(rec["frame"])
    cam = _camera(calib)
    raw = [{"index": int(i), "corners_cam0": c.tolist()} for i, c in zip(g["box_index_raw"], g["corners_cam0_raw"])]
    vis = pipeline.prepare_boxes(raw, cam, calib["TrVeloToCam"])
    assert [b["index"] for b in vis] == [int(g["box_index_raw"][p]) for p in g["visible_pos"]]
    got = np.array([b["corners_velo"] for b in vis]).reshape(-1, 8, 3)
    assert np.array_equal(got, g["corners_velo"])
    assert all(isinstance(b["corners_velo"], list) for b in vis)
    # the projected 2D boxes: same integers as cam2image on the host
    for b in vis:
        u, v, d = cam.cam2image(np.array(b["corners_cam0"]).T)
        f = d > 0
        assert b["_front"] == int(f.sum())
        if f.any():
            assert b["_bbox2d"] == [float(u[f].min()), float(v[f].min()), float(u[f].max()), float(v[f].max())]
    for kind in ("rect5", "edge"):
        boxes2d = g["boxes2d_" + kind]
        pairs = pipeline.match_detections_to_bboxes(boxes2d, vis, pipeline.default_colors(len(boxes2d)), cam)
        assert np.array_equal(np.array([p[0] for p in pairs]).reshape(-1, 8, 3), g["iou_match_corners_" + kind])


def test_depth_image_and_per_car_depth_maps(calib):
    """the last-writer depth scatter on the GPU == oracle, and the per-car maps == the reference loop."""
    g = load_golden(100)
    cam = _camera(calib)
    ctx = pipeline.get_context(0)
    ctx.set_camera(calib["TrVeloToRect"], cam.K, cam.width, cam.height, 0.0, 30.0)
    D, win = ctx.depth_image(g["points"])
    Dref, wref = orc.depth_image(g["points"], calib["TrVeloToRect"], calib["K"][:, :3], cam.width, cam.height, 0.0, 30.0)
    assert np.array_equal(D, Dref) and np.array_equal(win, wref)
    masks = unpack_masks(g, "rect5", cam.height, cam.width)
    maps = pipeline.per_car_depth_maps(g["points"], calib["TrVeloToRect"], cam, list(masks), depth_max=30.0)
    off = g["depthmap_off_rect5"]
    assert [cid for cid, _ in maps] == list(range(1, len(masks) + 1))
    for i, (_, dm) in enumerate(maps):
        flat = np.flatnonzero(dm)
        assert np.array_equal(flat, g["depthmap_idx_rect5"][off[i]:off[i + 1]])
        assert np.array_equal(dm.ravel()[flat], g["depthmap_val_rect5"][off[i]:off[i + 1]])
    # many points per pixel (collisions): a dense synthetic cloud
    from lidar_object_detection_amd import synthetic as S
    pts = S.synthetic_cloud(3_000_000, seed=9)
    ctx.set_camera(calib["TrVeloToRect"], cam.K, cam.width, cam.height, 0.0, 50.0)
    D, win = ctx.depth_image(pts)
    Dref, wref = orc.depth_image(pts, calib["TrVeloToRect"], calib["K"][:, :3], cam.width, cam.height, 0.0, 50.0)
    assert np.array_equal(D, Dref) and np.array_equal(win, wref)


def _dataset_tree(tmp_path, calib, frames_wanted):
    cam = _camera(calib)
    root = tmp_path / "KITTI360_sample"
    seq = "2013_05_28_drive_0000_sync"
    (root / "data_3d_raw" / seq / "velodyne_points" / "data").mkdir(parents=True)
    (root / "bboxes_3D_cam0").mkdir()
    (root / "data_2d_raw" / seq / "image_00" / "data_rect").mkdir(parents=True)
    gold = {}
    for r in golden_frames()["frames"]:
        if r["frame"] not in frames_wanted:
            continue
        g = load_golden(r["frame"])
        g["points"].tofile(str(root / "data_3d_raw" / seq / "velodyne_points" / "data" / ("%010d.bin" % r["frame"])))
        (root / "data_2d_raw" / seq / "image_00" / "data_rect" / ("%010d.png" % r["frame"])).write_bytes(b"")
        if "skipped" in r:
            continue
        raw = [{"index": int(i), "corners_cam0": c.tolist()} for i, c in zip(g["box_index_raw"], g["corners_cam0_raw"])]
        (root / "bboxes_3D_cam0" / ("BBoxes_%d.json" % r["frame"])).write_text(json.dumps(raw))
        gold[r["frame"]] = g
    velo = kitti360.Kitti360Viewer3DRaw(seq=0, root_dir=str(root))
    return root, seq, cam, velo, gold


def test_v3_v4_v5_entry_points(calib, tmp_path, monkeypatch):
    """process_frame_with_statistics (V3), process_frame (V4), projectVeloToImage (V5) on a rebuilt dataset tree."""
    root, seq, cam, velo, gold = _dataset_tree(tmp_path, calib, (100, 1461, 2717))
    monkeypatch.setattr(pipeline, "sequence_setup",
                        lambda path, s=0, c=0: (seq, cam, calib["TrVeloToCam"], calib["TrVeloToRect"], velo))

    def segmenter(image_path):
        frame = int(os.path.basename(image_path).split(".")[0])
        m = unpack_masks(gold[frame], "rect5", cam.height, cam.width)
        return None, m, pipeline.default_colors(len(m)), gold[frame]["boxes2d_rect5"], np.ones(len(m))

    seen = {}
    with contextlib.redirect_stdout(io.StringIO()) as out:
        res3 = pipeline.process_frame_with_statistics(0, 0, segmenter=segmenter, image_loader=lambda p: p, kitti360_path=str(root),
                                                      visualizer=lambda f, st, pv, bg: seen.setdefault(f, (st, pv, bg)))
    assert [r["frame"] for r in res3] == [100, 1461] and "SUMMARY STATISTICS" in out.getvalue()
    for r in res3:
        g = gold[r["frame"]]
        assert [d["points_inside_bbox"] for d in r["car_statistics"]] == g["stats_points_inside_bbox_rect5_d50"].tolist()
        st, pv, bg = seen[r["frame"]]
        assert len(pv) == len(g["valid_idx_d50"]) and bg.sum() == np.unpackbits(g["bg_assigned_rect5_d50"])[:len(pv)].sum()
    with contextlib.redirect_stdout(io.StringIO()):
        res4 = pipeline.process_frame(0, 0, segmenter=segmenter, image_loader=lambda p: p, kitti360_path=str(root))
        res5 = pipeline.projectVeloToImage(0, 0, segmenter=segmenter, image_loader=lambda p: p, kitti360_path=str(root))
    for r4, r5 in zip(res4, res5):
        g = gold[r4["frame"]]
        assert np.array_equal(r4["valid_indices"], g["valid_idx_d30"]) and np.array_equal(r5["valid_indices"], g["valid_idx_d30"])
        assert [len(s) for s in r4["car_point_sets"]] == g["inst_count_rect5_d30"].tolist()
        assert len(r4["remaining_points"]) == len(g["valid_idx_d30"]) - np.unpackbits(g["bg_assigned_rect5_d30"])[:len(g["valid_idx_d30"])].sum()
        assert np.array_equal(np.array([p[0] for p in r4["matched_pairs"]]).reshape(-1, 8, 3), g["iou_match_corners_rect5"])
        # V5 keeps every annotated box: all of them come back (matched in colour or grey)
        assert len(r5["matched_pairs"]) == len(g["corners_cam0_raw"])


def test_first_match_labelling_matches_same_color_loop(calib):
    """Same_color.py:113-131 (exclusive, first mask wins, mask > 0.5, depth < 30) on frame 100 with the
    nine overlapping 'edge' masks; golden = the reference's loop run by tests/golden/make_golden_views.py."""
    views = np.load(os.path.join(os.path.dirname(__file__), "golden", "views_golden.npz"))
    g = load_golden(100)
    cam = _camera(calib)
    masks = unpack_masks(g, "edge", cam.height, cam.width)
    masks[3] *= np.float32(0.75)                        # still > 0.5: a member here, not under astype(uint8)
    colors = pipeline.generate_consistent_colors(len(masks))
    r = pipeline.label_points_first_match(g["points"], calib["TrVeloToRect"], cam, masks, colors, depth_max=30)
    assert np.array_equal(r["car_idx"], views["samecolor_idx"])
    assert np.array_equal(r["car_mask"], views["samecolor_mask"].astype(np.int64))
    assert np.array_equal(r["background_idx"], views["samecolor_bg"])
    assert np.array_equal(r["colored_points"], g["points"][views["samecolor_idx"], :3])
    assert np.array_equal(r["full_points"], g["points"][views["samecolor_bg"], :3])
    assert np.array_equal(r["colored_colors"][:64], views["samecolor_colors"])
    assert len(set(views["samecolor_mask"].tolist())) > 3 and len(r["car_idx"]) + len(r["background_idx"]) == len(g["valid_idx_d30"])
    empty = pipeline.label_points_first_match(g["points"][:1000], calib["TrVeloToRect"], cam, [], None, depth_max=30)
    assert len(empty["car_idx"]) == 0 and len(empty["background_idx"]) == int(np.sum(g["valid_idx_d30"] < 1000))


def test_stream_frames_with_box_files_and_lazy_scan_gathers(calib, tmp_path):
    """stream_frames over three real frames with the reader's worker parsing the box files beside the scans (box_paths): the
    statistics are the goldens'; with gather=True every result outlives the reader, with gather=False the gathers from a scan are
    made on demand while that scan is the reader's current one and refused afterwards (its buffers hold another scan by then)."""
    from lidar_object_detection_amd._native import BOXES_PARSED, LpfError
    cam = _camera(calib)
    frames = (100, 250, 570)
    gs = {f: load_golden(f) for f in frames}
    scans, boxes = [], []
    for f in frames:
        g = gs[f]
        sp = tmp_path / ("%010d.bin" % f)
        g["points"].tofile(sp)
        bp = tmp_path / ("BBoxes_%d.json" % f)
        bp.write_text(json.dumps([{"index": int(i), "corners_cam0": c.tolist()} for i, c in zip(g["box_index_raw"], g["corners_cam0_raw"])]))
        scans.append(str(sp)); boxes.append(str(bp))
    seen = []

    def inputs_for(i, path, scan):
        f = frames[i]
        assert scan.boxes_state == BOXES_PARSED and scan.box_index.tolist() == gs[f]["box_index_raw"].tolist()
        assert np.array_equal(scan.boxes_cam0, gs[f]["corners_cam0_raw"])
        seen.append(f)
        m = unpack_masks(gs[f], "rect5", cam.height, cam.width)
        b = pipeline.prepare_boxes_from_arrays(scan.box_index, scan.boxes_cam0, cam, calib["TrVeloToCam"])
        return f, m, b, pipeline.default_colors(len(m))

    def stats_ok(r):
        g = gs[r["frame"]]
        for key in ("car_id", "matched_bbox_id", "total_points", "points_inside_bbox"):
            assert [d[key] for d in r["car_statistics"]] == g["stats_%s_rect5_d50" % key].tolist(), (r["frame"], key)

    kept = list(pipeline.stream_frames(scans, inputs_for, calib["TrVeloToRect"], cam, box_paths=boxes))
    assert seen == list(frames) and [r["frame"] for r in kept] == list(frames)
    for r in kept:                                                        # gathered before the reader moved on: good for ever
        stats_ok(r)
        g = gs[r["frame"]]
        assert np.array_equal(r["points_valid"], g["points"][g["valid_idx_d50"], :3])
        assert [len(c) for c in r["car_point_sets"]] == g["inst_count_rect5_d50"].tolist()
    prev = None
    for r in pipeline.stream_frames(scans, inputs_for, calib["TrVeloToRect"], cam, box_paths=boxes, gather=False):
        stats_ok(r)
        g = gs[r["frame"]]
        if r["frame"] != 250:                                             # read while the scan is the current one: the same gather
            assert np.array_equal(r["points_valid"], g["points"][g["valid_idx_d50"], :3])
        assert np.array_equal(r["bg_assigned"], np.unpackbits(g["bg_assigned_rect5_d50"])[:len(g["valid_idx_d50"])].astype(bool))   # (not a gather from the scan)
        if prev is not None and prev["frame"] == 250:                     # never read while live: refused now, not answered from another scan
            with pytest.raises(LpfError):
                prev["points_valid"]
            with pytest.raises(LpfError):
                prev["car_point_sets"]
        prev = r


def test_scan_reader_read_ahead(calib, tmp_path):
    """lpf_reader_*: files come back in submission order, bit-identical to np.fromfile (V3:24-28), the
    HBM copy gives the same results as the host path, errors name the file (V3:26-27) and do not stop
    the reader, and more files than buffers recycle the slots correctly."""
    from lidar_object_detection_amd._native import LpfError, ScanReader
    from lidar_object_detection_amd import synthetic as S
    _, T, K, W, H = S.default_calibration(calib)
    ctx = pipeline.get_context(0)
    ctx.set_camera(T, K, W, H, 0.0, 50.0)
    sc = S.scene(50000, n_masks=4, n_boxes=6, seed=5)
    ctx.set_masks(sc["masks"])
    ctx.set_boxes(sc["corners_velo"])
    sizes = [50000, 1, 0, 4096, 33333, 50000, 12345, 8, 40000]
    paths, clouds = [], []
    for i, n in enumerate(sizes):
        pts = S.synthetic_cloud(max(n, 1), seed=100 + i)[:n]
        p = tmp_path / ("%010d.bin" % i)
        pts.tofile(p)
        paths.append(p); clouds.append(pts)
    bad = tmp_path / "0000000099.bin"
    bad.write_bytes(b"\x00" * 20)                                   # not a multiple of 16 bytes
    missing = tmp_path / "0000000098.bin"
    order = paths[:3] + [missing] + paths[3:6] + [bad] + paths[6:]
    want = {str(p): c for p, c in zip(paths, clouds)}
    seen = 0
    with ScanReader(ctx, order, n_buffers=3, max_points=50000) as rd:
        assert len(rd) == len(order)
        it = iter(rd)
        prev = None
        for p in order:
            if p in (missing, bad):
                with pytest.raises(LpfError) as e:
                    next(it)
                assert str(p) in str(e.value) and (("does not exist!" in str(e.value)) == (p == missing))
                continue
            scan = next(it)
            assert scan.path == str(p) and scan.n == len(want[str(p)])
            assert np.array_equal(scan.points, np.fromfile(p, dtype=np.float32).reshape(-1, 4))
            r_dev = ctx.run(scan, want_float=True)                  # points taken from the HBM copy
            r_host = ctx.run(want[str(p)], want_float=True)         # staged from host memory
            for k in ("u", "v", "label_bits", "valid_idx", "depth", "count_mb", "best_box", "inst_count"):
                assert np.array_equal(r_dev[k], r_host[k], equal_nan=True), (p, k)
            for a, b in zip(r_dev["inst_lists"], r_host["inst_lists"]):
                assert np.array_equal(a, b)
            if prev is not None:
                with pytest.raises(LpfError, match="recycled"):
                    ctx.run(prev)
            prev = scan
            seen += 1
        with pytest.raises(StopIteration):
            next(it)
    assert seen == len(paths)
    with pytest.raises(LpfError, match="at most"):
        with ScanReader(ctx, [paths[0]], n_buffers=2, max_points=1000) as rd:
            next(iter(rd))
    with pytest.raises(LpfError, match="n_buffers"):
        ScanReader(ctx, [paths[0]], n_buffers=1)
    ctx.clear_masks(); ctx.clear_boxes()


def test_integration_md_ctypes_stub_runs(calib):
    """The raw ctypes stub printed in INTEGRATION.md section C is executed as written (frame 100, 5 masks, 25 boxes)
    and its results compared with the golden vectors: the document cannot drift from the ABI."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    sec = text[text.index("## C. Raw ctypes stub"):]
    code = re.search(r"```python\n(.*?)```", sec, re.S).group(1)
    code = code.replace('ctypes.CDLL("lidar_object_detection_amd/liblpf.so")',
                        'ctypes.CDLL(%r)' % os.path.join(root, "lidar_object_detection_amd", "liblpf.so"))
    g = load_golden(100)
    cam = _camera(calib)
    ns = {"TrVeloToRect": calib["TrVeloToRect"], "camera": cam, "points": np.ascontiguousarray(g["points"]),
          "masks": unpack_masks(g, "rect5", cam.height, cam.width), "m_": 2,
          "bboxes_3d": [{"corners_velo": c.tolist()} for c in g["corners_velo"]]}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    assert np.array_equal(ns["valid_indices"], g["valid_idx_d50"])
    M, B = 5, len(g["corners_velo"])
    assert np.array_equal(ns["cnt"].reshape(M, B), g["count_mb_rect5_d50"])
    off = np.concatenate([[0], np.cumsum(g["inst_count_rect5_d50"])])
    want = g["points"][g["inst_cat_rect5_d50"][off[2]:off[3]], :3]
    assert np.array_equal(ns["car_points_m"], want) and len(want) > 0
    assert np.array_equal(ns["uv"][:, 0], np.clip(g["u"], I32.min, I32.max))
    ns["lib"].lpf_destroy(ns["ctx"])


def test_integration_md_device_mode_snippet_runs(calib):
    """The device-mode snippet of INTEGRATION.md section C (explicit stream edges, masks that stay on the GPU), executed
    as written after the stub above it, on torch tensors; results against the golden vectors."""
    import ctypes
    import re
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    sec = text[text.index("## C. Raw ctypes stub"):]
    stub, dev_snip = re.findall(r"```python\n(.*?)```", sec, re.S)[:2]
    stub = stub.replace('ctypes.CDLL("lidar_object_detection_amd/liblpf.so")',
                        'ctypes.CDLL(%r)' % os.path.join(root, "lidar_object_detection_amd", "liblpf.so"))
    g = load_golden(100)
    cam = _camera(calib)
    ns = {"TrVeloToRect": calib["TrVeloToRect"], "camera": cam, "points": np.ascontiguousarray(g["points"]),
          "masks": unpack_masks(g, "rect5", cam.height, cam.width), "m_": 2,
          "bboxes_3d": [{"corners_velo": c.tolist()} for c in g["corners_velo"]]}
    exec(compile(stub, "INTEGRATION.md", "exec"), ns)            # context, camera, boxes (and a first, host-mode run)
    dev = torch.device("cuda", 0)
    n, M, B = len(g["points"]), 5, len(g["corners_velo"])
    d = dict(uv=torch.empty((n, 2), dtype=torch.int32, device=dev), lab=torch.empty(n, dtype=torch.int32, device=dev),
             vidx=torch.empty(n, dtype=torch.int64, device=dev), iidx=torch.empty(n, dtype=torch.int64, device=dev),
             cnt=torch.zeros(M * B, dtype=torch.int32, device=dev), summ=torch.zeros(928, dtype=torch.uint8, device=dev))
    o = ns["Outputs"](uv=d["uv"].data_ptr(), label_bits=d["lab"].data_ptr(), valid_idx=d["vidx"].data_ptr(), inst_idx=d["iidx"].data_ptr(),
                      inst_cap=n, count_mb=d["cnt"].data_ptr(), summary=d["summ"].data_ptr(), on_device=1)
    ns.update(torch=torch, o=o, n=n, M=M, pts=torch.from_numpy(ns["points"]).to(dev),
              result_masks=torch.from_numpy(ns["masks"]).to(dev))
    for f in ("lpf_set_stream", "lpf_wait_for_stream", "lpf_release_to_stream"):
        getattr(ns["lib"], f).argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    dev_snip = dev_snip.replace("lib.lpf_set_stream(ctx, P(cur))", "pass")   # the snippet shows both ways; take the explicit edges
    exec(compile(dev_snip, "INTEGRATION.md", "exec"), ns)
    nv = len(g["valid_idx_d50"])
    assert np.array_equal(d["vidx"].cpu().numpy()[:nv], g["valid_idx_d50"])    # .cpu() on torch's stream: behind the release edge
    assert np.array_equal(d["cnt"].cpu().numpy().reshape(M, B), g["count_mb_rect5_d50"])
    ns["lib"].lpf_destroy(ns["ctx"])


def test_integration_md_frame_job_snippet_runs(calib):
    """The one-call-per-frame snippet of INTEGRATION.md section C (lpf_frame_job / lpf_run_frame), executed as written after the stub:
    frame 100's scan, its masks with their rectangles, its annotated boxes as cam-0 corners; results against the golden vectors."""
    import ctypes
    import re
    import types
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    sec = text[text.index("## C. Raw ctypes stub"):]
    blocks = re.findall(r"```python\n(.*?)```", sec, re.S)
    stub, job_snip = blocks[0], [b for b in blocks if "lpf_run_frame" in b][0]
    stub = stub.replace('ctypes.CDLL("lidar_object_detection_amd/liblpf.so")',
                        'ctypes.CDLL(%r)' % os.path.join(root, "lidar_object_detection_amd", "liblpf.so"))
    g = load_golden(100)
    cam = _camera(calib)
    masks = unpack_masks(g, "rect5", cam.height, cam.width)
    ns = {"TrVeloToRect": calib["TrVeloToRect"], "camera": cam, "points": np.ascontiguousarray(g["points"]), "masks": masks, "m_": 2,
          "bboxes_3d": [{"corners_velo": c.tolist()} for c in g["corners_velo"]]}
    exec(compile(stub, "INTEGRATION.md", "exec"), ns)
    from lidar_object_detection_amd._native import LpfContext
    dev = torch.device("cuda", 0)
    n, M, B = len(g["points"]), 5, len(g["corners_cam0_raw"])
    d = dict(uv=torch.empty((n, 2), dtype=torch.int32, device=dev), lab=torch.empty(n, dtype=torch.int32, device=dev),
             vidx=torch.empty(n, dtype=torch.int64, device=dev), iidx=torch.empty(n, dtype=torch.int64, device=dev),
             cnt=torch.zeros(M * B, dtype=torch.int32, device=dev), summ=torch.zeros(928, dtype=torch.uint8, device=dev))
    out = ns["Outputs"](uv=d["uv"].data_ptr(), label_bits=d["lab"].data_ptr(), valid_idx=d["vidx"].data_ptr(), inst_idx=d["iidx"].data_ptr(),
                        inst_cap=n, count_mb=d["cnt"].data_ptr(), summary=d["summ"].data_ptr(), on_device=1)
    m8 = masks.astype(np.uint8)
    frame = types.SimpleNamespace(pts=torch.from_numpy(ns["points"]).to(dev), n=n, masks_u8=torch.from_numpy(m8).to(dev), M=M,
                                  rects_i32=torch.from_numpy(LpfContext.mask_rects(m8)).to(dev),
                                  corners_cam0=torch.from_numpy(np.ascontiguousarray(g["corners_cam0_raw"], dtype=np.float64)).to(dev), B=B, out=out)
    torch.cuda.synchronize(dev)
    ns.update(frame=frame, T_cam_to_velo=np.ascontiguousarray(np.linalg.inv(calib["TrVeloToCam"])))
    ns["lib"].lpf_run_frame.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    exec(compile(job_snip, "INTEGRATION.md", "exec"), ns)
    ns["lib"].lpf_sync.argtypes = [ctypes.c_void_p]
    assert ns["lib"].lpf_sync(ns["ctx"]) == 0
    nv = len(g["valid_idx_d50"])
    assert np.array_equal(d["vidx"].cpu().numpy()[:nv], g["valid_idx_d50"])
    got = d["cnt"].cpu().numpy().reshape(M, B)
    assert np.array_equal(got[:, g["visible_pos"]], g["count_mb_rect5_d50"])
    ns["lib"].lpf_destroy(ns["ctx"])


def test_integration_md_python_snippets_run(calib, tmp_path, monkeypatch):
    """Sections A and B of INTEGRATION.md executed as written on frame 100."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text[:text.index("## C. Raw ctypes stub")], re.S)
    assert len(blocks) == 3
    g = load_golden(100)
    cam = _camera(calib)
    monkeypatch.chdir(tmp_path)
    ns = {"TrVeloToRect": calib["TrVeloToRect"], "camera": cam, "points": np.ascontiguousarray(g["points"]), "frame": 100,
          "masks": unpack_masks(g, "rect5", cam.height, cam.width), "colors": pipeline.default_colors(5),
          "bboxes_3d": [{"corners_cam0": None, "corners_velo": c.tolist()} for c in g["corners_velo"]],
          "project_points": pipeline.project_points}
    with contextlib.redirect_stdout(io.StringIO()):
        for code in blocks:
            exec(compile(code, "INTEGRATION.md", "exec"), ns)
    assert np.array_equal(ns["valid_indices"], g["valid_idx_d50"]) and np.array_equal(ns["u_valid"], g["u"][g["valid_idx_d50"]])
    assert [d["points_inside_bbox"] for d in ns["car_statistics"]] == g["stats_points_inside_bbox_rect5_d50"].tolist()
    assert os.path.isfile(tmp_path / "results" / "master_car_statistics.csv")


def test_allreduce_metrics_over_rccl_single_rank():
    """lpf_allreduce_metrics with a real RCCL communicator (one rank = this GPU; the N-rank case is the same call):
    sum / min / max of an int64 vector come back unchanged, argument errors are reported."""
    import ctypes
    from lidar_object_detection_amd._native import LpfContext, LpfError
    rccl = None
    for name in ("librccl.so", "librccl.so.1"):
        try:
            rccl = ctypes.CDLL(name, mode=ctypes.RTLD_GLOBAL)
            break
        except OSError:
            continue
    if rccl is None:
        pytest.skip("librccl.so not found")
    comm = ctypes.c_void_p()
    dev = (ctypes.c_int * 1)(0)
    rccl.ncclCommInitAll.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    assert rccl.ncclCommInitAll(ctypes.byref(comm), 1, dev) == 0
    try:
        with LpfContext(0) as ctx:
            vec = np.array([3, 25, 21, 8833, 8274, 559, 7000, 412345], np.int64)     # the 8-word aggregate of distributed.py
            for op in ("sum", "min", "max"):
                got = ctx.allreduce_metrics(vec.copy(), comm, op)
                assert np.array_equal(got, vec), op
            big = np.arange(-5000, 5000, dtype=np.int64) * (1 << 40)
            assert np.array_equal(ctx.allreduce_metrics(big.copy(), comm), big)
            with pytest.raises(LpfError, match="allreduce_metrics"):
                ctx.allreduce_metrics(np.zeros(0, np.int64), comm)
            with pytest.raises(LpfError, match="comm"):
                ctx.allreduce_metrics(vec.copy(), ctypes.c_void_p(None))
    finally:
        rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        rccl.ncclCommDestroy(comm)


@pytest.mark.parametrize("rec", FRAMES, ids=lambda r: "frame%d" % r["frame"])
def test_boxes_from_cam0_corners_on_the_device(rec, calib):
    """V3:556-562 in one device-side step (lpf_set_boxes_cam0): the annotation's cam-0 corners go in, filter_visible_bboxes
    and transform_bboxes_to_velodyne happen on the GPU, the box tables are built there.  Kept boxes: same corners and same
    counts as the reference-generated golden vectors; dropped boxes: a zero column; best box: the same box."""
    g = load_golden(rec["frame"])
    if "corners_cam0_raw" not in g or "count_mb_rect5_d50" not in g:
        pytest.skip("no boxes for this frame")
    cam = _camera(calib)
    masks = unpack_masks(g, "rect5", cam.height, cam.width)
    M = len(masks)
    pos = g["visible_pos"]
    with LpfContext(0) as ctx:
        ctx.set_camera(calib["TrVeloToRect"], cam.K, cam.width, cam.height, 0.0, 50.0)
        ctx.set_masks(masks)
        (vis, cv, bb, fr), = ctx.set_boxes_cam0(g["corners_cam0_raw"], np.linalg.inv(calib["TrVeloToCam"]))
        assert np.array_equal(np.flatnonzero(vis), pos)
        assert np.array_equal(cv[pos], g["corners_velo"])
        r = ctx.run(g["points"])
        cm = r["count_mb"]
        assert cm.shape == (M, len(vis))
        assert np.array_equal(cm[:, pos], g["count_mb_rect5_d50"]) and not cm[:, ~vis].any()
        # the same boxes, already filtered, through the host-corner path: the best-box scan must name the same boxes
        ctx.set_boxes(g["corners_velo"])
        r2 = ctx.run(g["points"])
        assert np.array_equal(r2["count_mb"], g["count_mb_rect5_d50"])
        filtered = np.cumsum(vis) - 1                        # position in the reference's filtered list
        for m in range(M):
            if r2["best_box"][m] >= 0:
                assert filtered[r["best_box"][m]] == r2["best_box"][m] and vis[r["best_box"][m]]
            else:
                assert r["best_box"][m] == -1
            assert r["best_cnt"][m] == r2["best_cnt"][m]


def test_hipgraph_with_boxes_that_change_every_replay(calib):
    """The per-frame loop of the reference prepares a new box list for every frame (V3:556-562).  Captured once -- box
    set-up from a device buffer, mask pack, the whole hot path -- and replayed on new points, masks AND boxes."""
    import torch
    from lidar_object_detection_amd import synthetic as S
    from lidar_object_detection_amd._native import SUMMARY_DTYPE
    _, T, K, W, H = S.default_calibration(calib)
    dev = torch.device("cuda", 0)
    n, M, Bx = 120_000, 5, 21
    stream = torch.cuda.Stream(dev)
    with torch.cuda.stream(stream), LpfContext(0) as ctx:
        ctx.set_stream(stream.cuda_stream)
        ctx.set_camera(T, K, W, H, 0.0, 50.0)
        sc0 = S.scene(n, n_masks=M, n_boxes=Bx, seed=900)
        pts = torch.from_numpy(sc0["points"]).to(dev)
        masks = torch.from_numpy(sc0["masks"]).to(dev)
        corners = torch.from_numpy(np.ascontiguousarray(sc0["corners_velo"])).to(dev).contiguous()
        off = np.array([0, n], np.int64)
        boff = np.array([0, Bx], np.int32)
        o = dict(uv=torch.empty((n, 2), dtype=torch.int32, device=dev), label_bits=torch.empty(n, dtype=torch.int32, device=dev),
                 valid_idx=torch.empty(n, dtype=torch.int64, device=dev), inst_idx=torch.empty(n, dtype=torch.int64, device=dev),
                 count_mb=torch.zeros(M * Bx, dtype=torch.int32, device=dev),
                 summary=torch.zeros(SUMMARY_DTYPE.itemsize, dtype=torch.uint8, device=dev))
        step = ctx.make_device_step(pts, off, masks_u8=masks.unsqueeze(0), erode_iters=1, inst_cap=n, **o)

        def frame():
            ctx.set_boxes_device(corners, boff)
            step()

        frame()                                             # warm: allocations + table uploads happen here
        ctx.sync()
        ctx.graph_begin()
        frame()
        g = ctx.graph_end()
        for k in range(3):
            sc = S.scene(n, n_masks=M, n_boxes=Bx, seed=901 + k)
            pts.copy_(torch.from_numpy(sc["points"]))
            masks.copy_(torch.from_numpy(sc["masks"]))
            corners.copy_(torch.from_numpy(np.ascontiguousarray(sc["corners_velo"])))
            stream.synchronize()
            ctx.graph_launch(g)
            ctx.sync()
            sm = np.frombuffer(o["summary"].cpu().numpy().tobytes(), SUMMARY_DTYPE)[0]
            ref = orc.run(sc["points"], T, K, W, H, 0.0, 50.0, label_img=orc.pack_masks(sc["masks"], 1, H, W), M=M,
                          corners=sc["corners_velo"], want_float=False)
            assert int(sm["n_valid"]) == ref["n_valid"] and np.array_equal(sm["inst_count"][:M], ref["inst_count"])
            assert np.array_equal(o["count_mb"].cpu().numpy().reshape(M, Bx), ref["count_mb"]) and int(ref["count_mb"].sum()) > 0
            assert np.array_equal(sm["best_box"][:M], ref["best_box"]) and np.array_equal(sm["best_cnt"][:M], ref["best_cnt"])
        ctx.graph_destroy(g)


def test_run_frames_with_masks_that_stay_on_the_gpu(calib):
    """run_frames given YOLO's mask tensor as it is on the GPU (float32, no .cpu().numpy()) returns what it returns for
    the same masks as host arrays -- the reference-generated golden statistics of frame 100."""
    import torch
    g = load_golden(100)
    cam = _camera(calib)
    masks = unpack_masks(g, "rect5", cam.height, cam.width)                # float32 [5,H,W]
    boxes = [{"corners_velo": c.tolist()} for c in g["corners_velo"]]
    dev = torch.device("cuda", 0)
    a = pipeline.run_frames([pipeline.FrameInputs(100, g["points"], masks, boxes)], calib["TrVeloToRect"], cam, 50.0, 10, True)[0]
    b = pipeline.run_frames([pipeline.FrameInputs(100, g["points"], torch.from_numpy(masks).to(dev), boxes)], calib["TrVeloToRect"],
                            cam, 50.0, 10, True)[0]
    assert [d["points_inside_bbox"] for d in b["car_statistics"]] == g["stats_points_inside_bbox_rect5_d50"].tolist()
    assert np.array_equal(a["count_mb"], b["count_mb"]) and np.array_equal(a["valid_indices"], b["valid_indices"])
    assert all(np.array_equal(x, y) for x, y in zip(a["car_point_sets"], b["car_point_sets"]))
    assert np.array_equal(a["bg_assigned"], b["bg_assigned"])


@pytest.mark.parametrize("order", ["liblpf_first", "torch_first"])
def test_both_load_orders_work_on_the_gpu(order):
    """Raw ctypes user of the C ABI and PyTorch in one process, liblpf.so loaded before or after torch: both must then be able
    to use the GPU (one HIP runtime, see tests/test_abi_exports.py)."""
    import subprocess
    import sys
    from lidar_object_detection_amd import _native
    lib = _native.library_path()
    load_lpf = 'import ctypes; lib = ctypes.CDLL("%s")' % lib
    use = ('ctx = ctypes.c_void_p(); lib.lpf_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int]\n'
           'rc = lib.lpf_create(ctypes.byref(ctx), 0); assert rc == 0, rc\n'
           'assert torch.cuda.is_available(); x = torch.arange(8, device="cuda").sum().item(); assert x == 28\n'
           'lib.lpf_sync.argtypes = [ctypes.c_void_p]; assert lib.lpf_sync(ctx) == 0\n'
           'lib.lpf_destroy.argtypes = [ctypes.c_void_p]; lib.lpf_destroy(ctx); print("OK")')
    code = (load_lpf + "\nimport torch\n" if order == "liblpf_first" else "import torch\n" + load_lpf + "\n") + use
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])


def test_more_than_32_detections_in_a_frame(calib):
    """The reference loops over every mask (V3:220), however many: run_frames takes 40 detections in two passes of the kernels
    (32 + 8) and returns what one unbounded pass would -- per-detection point sets, counts and statistics in detection order."""
    from lidar_object_detection_amd import synthetic as S
    _, T, K, W, H = S.default_calibration(calib)
    cam = _camera(calib)
    sc = S.scene(90_000, n_masks=40, n_boxes=12, seed=4242, calib=calib)
    masks = sc["masks"].astype(np.float32)
    boxes = [{"corners_velo": c.tolist()} for c in sc["corners_velo"]]
    r = pipeline.run_frames([pipeline.FrameInputs(7, sc["points"], masks, boxes)], calib["TrVeloToRect"], cam, 50.0, 10, True)[0]
    assert len(r["car_point_sets"]) == 40 and r["count_mb"].shape == (40, 12)
    any_mask = np.zeros(len(r["valid_indices"]), bool)
    for g0 in (0, 32):
        grp = sc["masks"][g0:g0 + 32]
        o = orc.run(sc["points"], T, K, W, H, 0.0, 50.0, label_img=orc.pack_masks(grp, 0, H, W), M=len(grp), corners=sc["corners_velo"],
                    want_float=False)
        assert np.array_equal(r["valid_indices"], o["valid_idx"])
        assert np.array_equal(r["count_mb"][g0:g0 + len(grp)], o["count_mb"])
        for m in range(len(grp)):
            assert np.array_equal(r["car_point_sets"][g0 + m], sc["points"][o["inst_lists"][m], :3])
        any_mask |= o["label_bits"][o["valid_idx"]] != 0
    assert np.array_equal(r["bg_assigned"], any_mask)
    ids = [d["car_id"] for d in r["car_statistics"]]
    assert ids == sorted(ids) and all(0 <= i < 40 for i in ids) and len(set(ids)) == len(ids)
    assert [d["total_points"] for d in r["car_statistics"]] == [len(r["car_point_sets"][i]) for i in ids]


def test_integration_md_frame_loop_with_boxes_per_frame_runs(calib):
    """The pipelined frame loop of INTEGRATION.md (masks and cam-0 boxes per frame, lent; lpf_set_pipelined(4)) executed as
    written over three real frames, through raw ctypes; counts against the golden vectors, and lpf_get_stats shows that the
    loop neither waited nor drained."""
    import ctypes
    import re
    import types
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    sec = text[text.index("## C. Raw ctypes stub"):]
    blocks = re.findall(r"```python\n(.*?)```", sec, re.S)
    stub, loop = blocks[0], blocks[2]
    assert "lpf_set_boxes_cam0" in loop and "lpf_set_pipelined" in loop
    stub = stub.replace('ctypes.CDLL("lidar_object_detection_amd/liblpf.so")',
                        'ctypes.CDLL(%r)' % os.path.join(root, "lidar_object_detection_amd", "liblpf.so"))
    g0 = load_golden(100)
    cam = _camera(calib)
    ns = {"TrVeloToRect": calib["TrVeloToRect"], "camera": cam, "points": np.ascontiguousarray(g0["points"]),
          "masks": unpack_masks(g0, "rect5", cam.height, cam.width), "m_": 2,
          "bboxes_3d": [{"corners_velo": c.tolist()} for c in g0["corners_velo"]]}
    exec(compile(stub, "INTEGRATION.md", "exec"), ns)            # context, camera (and a first, host-mode run)
    dev = torch.device("cuda", 0)
    frames, keep = [], []
    for fr in (100, 250, 360) * 2:                               # twice: the second pass finds every buffer at size
        g = load_golden(fr)
        mk = unpack_masks(g, "rect5", cam.height, cam.width).astype(np.uint8)
        n, M, B = len(g["points"]), len(mk), len(g["corners_cam0_raw"])
        t = dict(pts=torch.from_numpy(np.ascontiguousarray(g["points"], dtype=np.float32)).to(dev),
                 masks=torch.from_numpy(mk).to(dev),
                 cam0=torch.from_numpy(np.ascontiguousarray(g["corners_cam0_raw"], dtype=np.float64)).to(dev),
                 uv=torch.empty((n, 2), dtype=torch.int32, device=dev), lab=torch.empty(n, dtype=torch.int32, device=dev),
                 vidx=torch.empty(n, dtype=torch.int64, device=dev), iidx=torch.empty(n, dtype=torch.int64, device=dev),
                 cnt=torch.zeros(M * B, dtype=torch.int32, device=dev), summ=torch.zeros(928, dtype=torch.uint8, device=dev))
        out = ns["Outputs"](uv=t["uv"].data_ptr(), label_bits=t["lab"].data_ptr(), valid_idx=t["vidx"].data_ptr(), inst_idx=t["iidx"].data_ptr(),
                            inst_cap=n, count_mb=t["cnt"].data_ptr(), summary=t["summ"].data_ptr(), on_device=1)
        frames.append(types.SimpleNamespace(masks_u8=t["masks"], M=M, corners_cam0=t["cam0"], box_off=np.array([0, B], np.int32), pts=t["pts"],
                                            n=n, out=out, g=g, t=t, B=B))
        keep.append(t)
    torch.cuda.synchronize(dev)
    lib, ctx = ns["lib"], ns["ctx"]
    lib.lpf_get_stats.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    ns.update(frames=frames[:3], T_cam_to_velo=np.ascontiguousarray(np.linalg.inv(calib["TrVeloToCam"])), torch=torch)
    exec(compile(loop, "INTEGRATION.md", "exec"), ns)            # warm pass
    st = np.zeros(8, np.int64)
    lib.lpf_get_stats(ctx, st.ctypes.data, 8, 1)
    ns["frames"] = frames[3:]
    counted = loop.replace("lib.lpf_set_pipelined(ctx, 4)", "lib.lpf_set_pipelined(ctx, 4); lib.lpf_get_stats(ctx, st.ctypes.data, 8, 1)", 1)
    counted = counted.replace("lib.lpf_sync(ctx)", "lib.lpf_get_stats(ctx, st.ctypes.data, 8, 0); lib.lpf_sync(ctx)")     # counters of the loop itself
    exec(compile(counted, "INTEGRATION.md", "exec"), dict(ns, st=st))
    assert st[0] == 0 and st[1] == 0 and st[5] == 3, st        # no host wait, no drain, three riding box jobs
    for f in frames[3:]:
        cm = f.t["cnt"].cpu().numpy().reshape(f.M, f.B)
        pos = f.g["visible_pos"]
        rest = np.ones(f.B, bool); rest[pos] = False
        assert np.array_equal(cm[:, pos], f.g["count_mb_rect5_d50"]) and not cm[:, rest].any()
        sm = np.frombuffer(f.t["summ"].cpu().numpy().tobytes(), SUMMARY_DTYPE)[0]
        nv = len(f.g["valid_idx_d50"])
        assert int(sm["n_valid"]) == nv and np.array_equal(f.t["vidx"][:nv].cpu().numpy(), f.g["valid_idx_d50"])
    lib.lpf_destroy(ctx)
