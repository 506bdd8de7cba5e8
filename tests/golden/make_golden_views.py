#!/usr/bin/env python3
"""Golden vectors for the box-view filter, the detailed box projection, the colour table and the
exclusive (first-match) point labelling -- made by running the REFERENCE's own functions
(secondtest.py, V5_ProjectingBBoxes.py, firsttest.py) in the build container.  Same rules as
make_golden.py: modules are imported in place behind inert stubs, only inputs/outputs are written.

  * is_bbox_in_camera_view / filter_bboxes_in_camera_view   secondtest.py:277-419
  * project_3d_bbox_to_2d                                   V5:215-252 (detailed), firsttest.py:172-193 (plain)
  * generate_consistent_colors                              V5:88-121
  * match_detections_to_bboxes(iou_threshold=0.1)           firsttest.py:218-260
  * the exclusive first-match loop of Same_color.py:113-131 (inline in its main loop, so it is executed
    here as the same Python statements, on frame 100 with the overlapping "edge" masks of make_golden.py)

Usage: python tests/golden/make_golden_views.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402

REASONS = ("valid", "no_corners", "all_behind_camera", "no_intersection", "too_small", "error")


def main():
    G._seed_import_stubs()
    second = G._load_ref("secondtest.py", "ref_second")
    v5 = G._load_ref("V5_ProjectingBBoxes.py", "ref_v5b")
    first = G._load_ref("firsttest.py", "ref_first")
    v3 = G._load_ref("V3_point_cloud_with_erosion.py", "ref_v3b")
    kitti360 = G.kitti360
    camera = kitti360.CameraPerspective(G.DATA, G.SEQ, 0)
    velo_to_cam, velo_to_rect = kitti360.velo_to_rect_transforms(G.DATA, camera, 0)
    frames = kitti360.Kitti360Viewer3DRaw(seq=0, root_dir=G.DATA).available_frames()
    out = {"frames": [], "box_off": [0]}
    keep, reason, in_view, valid_depth, avg_depth = [], [], [], [], []
    p_ok, p_bbox, p_center, p_size, p_area, p_avg = [], [], [], [], [], []
    kept_count = []
    m_off, m_corners, m_color = [0], [], []
    for frame in frames:
        raw = G._quiet(v3.load_bounding_boxes, os.path.join(G.DATA, "bboxes_3D_cam0", "BBoxes_%d.json" % frame))
        if not raw:
            continue
        out["frames"].append(frame)
        out["box_off"].append(out["box_off"][-1] + len(raw))
        for b in raw:
            ok, info = second.is_bbox_in_camera_view(b, camera)
            keep.append(bool(ok)); reason.append(REASONS.index(info["reason"]))
            in_view.append(int(info.get("corners_in_view", -1)))
            valid_depth.append(int(info.get("corners_with_valid_depth", -1)))
            avg_depth.append(float(info.get("avg_depth", np.nan)))
            pi, _ = G._quiet(v5.project_3d_bbox_to_2d, b, camera)
            p_ok.append(pi is not None)
            if pi is None:
                pi = {"bbox": [0] * 4, "center": [0] * 2, "size": [0] * 2, "area": 0, "avg_depth": np.nan}
            p_bbox.append(pi["bbox"]); p_center.append(pi["center"]); p_size.append(pi["size"])
            p_area.append(pi["area"]); p_avg.append(pi["avg_depth"])
        filt, stats = G._quiet(second.filter_bboxes_in_camera_view, raw, camera, False)
        kept_count.append(stats["kept"])
        assert [b["index"] for b in filt] == [b["index"] for b, k in zip(raw, keep[-len(raw):]) if k]
        # firsttest IoU match (threshold 0.1) against shifted copies of the first kept boxes' projections
        boxes3d = G._quiet(first.transform_bboxes_to_velodyne, [dict(b) for b in raw], velo_to_cam)
        dets = []
        for b in filt[:6]:
            bb, _ = first.project_3d_bbox_to_2d(b, camera)
            dets.append([bb[0] + 3.0, bb[1] - 2.0, bb[2] + 17.0, bb[3] + 5.0])
        dets.append([5000.0, 5000.0, 5100.0, 5100.0])                        # matches nothing
        colors = v5.generate_consistent_colors(len(dets))
        pairs = G._quiet(first.match_detections_to_bboxes, np.array(dets, np.float32), boxes3d, colors, camera)
        m_off.append(m_off[-1] + len(pairs))
        m_corners += [np.asarray(p[0], np.float64) for p in pairs]
        m_color += [np.asarray(p[1], np.float64) for p in pairs]
    res = dict(frames=np.array(out["frames"], np.int64), box_off=np.array(out["box_off"], np.int64),
               keep=np.array(keep, bool), reason=np.array(reason, np.int8), corners_in_view=np.array(in_view, np.int64),
               corners_with_valid_depth=np.array(valid_depth, np.int64), avg_depth=np.array(avg_depth, np.float64),
               kept_count=np.array(kept_count, np.int64),
               proj_ok=np.array(p_ok, bool), proj_bbox=np.array(p_bbox, np.int64), proj_center=np.array(p_center, np.float64),
               proj_size=np.array(p_size, np.int64), proj_area=np.array(p_area, np.int64), proj_avg_depth=np.array(p_avg, np.float64),
               first_match_off=np.array(m_off, np.int64), first_match_corners=np.array(m_corners, np.float64).reshape(-1, 8, 3),
               first_match_color=np.array(m_color, np.float64).reshape(-1, 3),
               colors40=np.array(v5.generate_consistent_colors(40), np.int64))

    # Same_color.py:114-132 on frame 100: the main loop's OWN source lines, sliced from the file and executed (G.run_ref).
    # The loop collects points and colours, not indices: it is given points whose coordinates ARE their index and a
    # palette whose colour i IS i, so both are read back from what it collected.
    z = np.load(os.path.join(HERE, "frame_0000000100.npz"))
    points = z["points"]
    masks = np.unpackbits(z["masks_edge_packed"], axis=2)[:, :, :camera.width].astype(np.float32)
    mask_colors = v5.generate_consistent_colors(len(masks))
    ns = G.run_ref("V3_point_cloud_with_erosion.py", 565, 569, {"np": np, "points": points, "TrVeloToRect": velo_to_rect, "camera": camera},
                   "points_homo = points.copy()")
    tagged = np.repeat(np.arange(len(points), dtype=np.float64)[:, None], 4, axis=1)
    ns = G.run_ref("Same_color.py", 114, 132, {"np": np, "u": ns["u"], "v": ns["v"], "depth": ns["depth"], "camera": camera, "frame": 100,
                                                "points": tagged, "masks": list(masks), "mask_colors": [(i, i, i) for i in range(len(masks))]},
                   "valid = (u >= 0)")
    col_idx = [int(p[0]) for p in ns["colored_points"]]
    col_mask = [int(round(float(c[0]) * 255.0)) for c in ns["colored_colors"]]
    bg_idx = [int(p[0]) for p in ns["full_points"]]
    res.update(samecolor_idx=np.array(col_idx, np.int64), samecolor_mask=np.array(col_mask, np.int8),
               samecolor_bg=np.array(bg_idx, np.int64),
               samecolor_colors=np.array([np.array(mask_colors[i]) / 255.0 for i in col_mask[:64]], np.float64))
    np.savez_compressed(os.path.join(HERE, "views_golden.npz"), **res)
    print("boxes", len(keep), "kept", int(np.sum(keep)), "reasons", np.bincount(reason, minlength=6).tolist(),
          "first-match pairs", m_off[-1], "samecolor", len(col_idx), len(bg_idx))


if __name__ == "__main__":
    main()
