"""The reference's function surface for the hot path, served by the HIP library.

Names, argument meaning, return types and print formats follow
``Coding_testes/V3_point_cloud_with_erosion.py`` / ``cvs_erosion.py`` /
``V4_BBox_IoU_filtering.py`` (paths relative to /root/reference; line numbers cited per
function).  Everything that touches per-point data goes through ``liblpf.so``
(``_native.LpfContext``); what stays on the host is box-level scalar logic (8 corners per
box), dict assembly, printing and CSV -- none of it is a fallback for the kernels, and
there is no CPU path for them: without the GPU library these functions raise.

YOLO segmentation and Open3D stay outside (reference: unchanged subsystems); the entry
points take them as callables.
"""
import os
from datetime import datetime

import numpy as np

from . import _native, kitti360
from ._native import LpfContext, LPF_MAX_MASKS, Scan, ScanReader

_CONTEXTS = {}


def get_context(device=0):
    """Process-wide LpfContext of one GPU (created on first use)."""
    ctx = _CONTEXTS.get(device)
    if ctx is None:
        ctx = _CONTEXTS[device] = LpfContext(device)
    return ctx


def _f32_points(points, what="points"):
    """float32 view of caller points; refuses values a float32 cannot hold (the kernels
    take the velodyne float32 format, V3:28, and there is no float64 point path)."""
    p = np.asarray(points)
    if p.dtype == np.float32:
        return np.ascontiguousarray(p)
    q = np.ascontiguousarray(p, dtype=np.float32)
    if not np.array_equal(q.astype(np.float64), np.asarray(p, dtype=np.float64), equal_nan=True):
        raise TypeError("%s must hold float32-representable values (velodyne .bin format)" % what)
    return q


def _is_device_tensor(x):
    return type(x).__module__.startswith("torch") and getattr(x, "is_cuda", False)


def _mask_stack(masks, camera, resize_ctx=None, erode_iters=0, v3_pipeline=False, force_chain=False):
    """([M,H,W] float32/uint8 array from the reference's mask list, eroded_already).  A torch tensor that is already on the GPU --
    ``result.masks.data`` before the reference's ``.cpu().numpy()`` (V3:72) -- is passed through: the kernels read it where it is.
    Masks that do not arrive at the camera's size (the reference's scripts pass retina_masks=True, V3:64, so theirs do) go through
    ``cv2.resize(mask.astype(np.uint8), (W, H))`` as V3:222 does it -- on the GPU, by ``resize_ctx`` (whose camera must have been
    set; LpfContext.resize_masks) -- and come back as uint8 [M,H,W], nonzero = the reference's ``> 0.5``.  With the V3 erosion block
    in force (``v3_pipeline`` / ``erode_iters``) such masks take V3's own order: ``(mask * 255).astype(uint8)`` -> ``cv2.erode`` AT THE
    MASKS' OWN SIZE -> ``/ 255.0`` (V3:82-97), and only then ``astype(uint8)`` + resize (V3:222) -- all on the GPU
    (LpfContext.erode_masks, then resize_masks); the second value tells the caller that the erosion has been done.  ``force_chain``:
    masks at camera size take the same explicit chain (a batch is eroded either all inside lpf_set_masks_* or all here)."""
    def chain(m):
        """off-size masks -> uint8 [M,H,W] at camera size (device tensor in -> device tensor out)"""
        if resize_ctx is None:
            raise NotImplementedError("masks must already be %dx%d (retina_masks=True, V3:64) in this call" % (camera.height, camera.width))
        if not (erode_iters or v3_pipeline):
            return resize_ctx.resize_masks(m), False
        dev = _is_device_tensor(m)
        if dev:
            import torch
            f = m.to(torch.float32)
            u8 = (f * 255).to(torch.uint8) if v3_pipeline else f.to(torch.uint8)       # (mask * 255).astype(np.uint8), V3:85
            er = resize_ctx.erode_masks(u8.contiguous(), erode_iters)                 # cv2.erode at the masks' own size, V3:86-90
            back = (er == 255).to(torch.uint8) if v3_pipeline else er                 # / 255.0 (V3:93) then astype(np.uint8) (V3:222): 1 only for 255
        else:
            f = np.asarray(m)
            u8 = (f.astype(np.float32) * 255).astype(np.uint8) if v3_pipeline else f.astype(np.uint8)
            er = resize_ctx.erode_masks(u8, erode_iters)
            back = (er.astype(np.float32) / 255.0).astype(np.uint8) if v3_pipeline else er
        return resize_ctx.resize_masks(back), True
    if _is_device_tensor(masks):
        import torch
        if masks.ndim != 3:
            raise ValueError("device masks must be [M,h,w]")
        if tuple(masks.shape[1:]) != (camera.height, camera.width) or (force_chain and masks.shape[0]):
            return chain(masks)
        if masks.dtype == torch.bool:
            masks = masks.to(torch.uint8)
        elif masks.dtype not in (torch.float32, torch.uint8):
            masks = masks.to(torch.float32)
        return masks.contiguous(), False
    m = np.asarray(masks)
    if m.size == 0:
        return np.zeros((0, camera.height, camera.width), np.uint8), False
    if m.ndim != 3:
        raise ValueError("masks must be [M,h,w]")
    if m.shape[1:] != (camera.height, camera.width) or force_chain:
        return chain(m)
    if m.dtype.kind == "f":
        return np.ascontiguousarray(m, dtype=np.float32), False
    return np.ascontiguousarray(m.astype(np.uint8)), False


# ---------------------------------------------------------------------------------------
# box preparation (host scalars: 8 corners per box)
# ---------------------------------------------------------------------------------------
def filter_visible_bboxes(bboxes_3d, camera):
    """Keep boxes with >= 2 corners in front (depth > 0.1) and inside the image (V3:121-140).
    As in the reference the corners are projected without R_rect."""
    filtered = []
    for bbox in bboxes_3d:
        if "corners_cam0" not in bbox:
            continue
        corners = np.array(bbox["corners_cam0"])
        u, v, depth = camera.cam2image(corners.T)
        ok = (depth > 0.1) & (u >= 0) & (u < camera.width) & (v >= 0) & (v < camera.height)
        if np.sum(ok) >= 2:
            filtered.append(bbox)
    return filtered


def transform_bboxes_to_velodyne(bboxes_3d, TrVeloToCam):
    """Adds 'corners_velo' (list) to every box dict, in place (V3:41-52)."""
    cam_to_velo = np.linalg.inv(TrVeloToCam)
    for bbox in bboxes_3d:
        if "corners_cam0" in bbox:
            c = np.array(bbox["corners_cam0"])
            homo = np.hstack([c, np.ones((c.shape[0], 1))])
            bbox["corners_velo"] = np.matmul(cam_to_velo, homo.T).T[:, :3].tolist()
    return bboxes_3d


def prepare_boxes(bboxes_3d_raw, camera, TrVeloToCam, device=0, keep_all=False, as_arrays=False):
    """filter_visible_bboxes + transform_bboxes_to_velodyne (+ V4's projected 2D box) in one GPU call
    (SURVEY 8f-1): returns the visible boxes, in order, each a copy of the input dict with
    'corners_velo' (list, as the reference stores it) and '_bbox2d' / '_front' for the IoU match.
    The camera of the context is set to (camera.K, width, height).  ``keep_all=True`` skips the
    visibility filter (V5 transforms every box, V5:454-461).  ``as_arrays=True`` (callers that keep the boxes to themselves, like
    process_frames, whose only product is the CSV) stores 'corners_velo' as the f64 [8,3] array instead of the reference's
    ``.tolist()`` of it -- the same values; 7 500 Python floats per frame of 314 boxes that nobody reads cost more than the frame's kernels."""
    have = [b for b in bboxes_3d_raw if "corners_cam0" in b]
    if not have:
        return []
    ctx = get_context(device)
    ctx.ensure_intrinsics(camera.K, camera.width, camera.height)
    corners = np.array([b["corners_cam0"] for b in have], np.float64).reshape(-1, 8, 3)
    vis, cv, bb, fr = ctx.prepare_boxes(corners, np.linalg.inv(TrVeloToCam))
    out = _PreparedBoxes()
    bb_l, fr_l, vis_l = bb.tolist(), fr.tolist(), vis.tolist()      # (one conversion each: per box it cost more than the kernels)
    cv_l = cv if as_arrays else cv.tolist()
    for i, b in enumerate(have):
        if vis_l[i] or keep_all:
            d = dict(b)
            d["corners_velo"] = cv_l[i]
            d["_cv"] = cv[i]                                 # the same values as an array (private: spares run_frames the way back from lists)
            d["_bbox2d"] = bb_l[i] if fr_l[i] > 0 else None
            d["_front"] = fr_l[i]
            out.append(d)
    out.cv = cv if keep_all else cv[vis]
    return out


class _PreparedBoxes(list):
    """prepare_boxes' list of box dicts, with the corners of all of them as ONE float64 [B,8,3] array beside it (``cv``: what
    run_frames hands to the GPU, without a walk over the dicts).  A list in every other respect; ``cv`` is dropped by anything that
    makes a new list of it (slices, filters), and ``_corners_velo`` only trusts it while the lengths agree."""
    __slots__ = ("cv",)

    def __init__(self, *a):
        super().__init__(*a)
        self.cv = None


def prepare_boxes_from_arrays(index, corners_cam0, camera, TrVeloToCam, device=0, keep_all=False):
    """prepare_boxes for a box file the library has parsed (lpf_parse_boxes_json / the read-ahead reader: ``index`` int32 [B],
    ``corners_cam0`` float64 [B,8,3] -- the doubles json.load gives): the visible boxes, in order, as dicts with 'index',
    'corners_velo' (the f64 [8,3] array, as prepare_boxes(as_arrays=True) stores it), '_bbox2d' / '_front'.  The cam-0 corners are
    not copied into per-box lists: this is the form for callers that keep the boxes to themselves (process_frames)."""
    corners = np.ascontiguousarray(corners_cam0, dtype=np.float64).reshape(-1, 8, 3)
    out = _PreparedBoxes()
    if not len(corners):
        return out
    ctx = get_context(device)
    ctx.ensure_intrinsics(camera.K, camera.width, camera.height)
    vis, cv, bb, fr = ctx.prepare_boxes(corners, np.linalg.inv(TrVeloToCam))
    keep = np.arange(len(corners)) if keep_all else np.flatnonzero(vis)
    idx_l, bb_l, fr_l = np.asarray(index)[keep].tolist(), bb[keep].tolist(), fr[keep].tolist()
    out.cv = cv[keep]
    for j in range(len(keep)):
        c = out.cv[j]
        out.append({"index": idx_l[j], "corners_velo": c, "_cv": c, "_bbox2d": bb_l[j] if fr_l[j] > 0 else None, "_front": fr_l[j]})
    return out


def _corners_velo(bboxes_3d):
    """f64 [B,8,3] of the boxes that carry 'corners_velo' + their positions in the list."""
    if isinstance(bboxes_3d, _PreparedBoxes) and bboxes_3d.cv is not None and len(bboxes_3d.cv) == len(bboxes_3d):
        return bboxes_3d.cv, list(range(len(bboxes_3d)))     # (every dict of prepare_boxes carries 'corners_velo')
    pos = [i for i, b in enumerate(bboxes_3d) if "corners_velo" in b]
    if not pos:
        return np.zeros((0, 8, 3)), pos
    if all("_cv" in bboxes_3d[i] for i in pos):             # boxes that come from prepare_boxes carry their corners as arrays too
        return np.stack([bboxes_3d[i]["_cv"] for i in pos]).reshape(-1, 8, 3), pos
    return np.array([bboxes_3d[i]["corners_velo"] for i in pos], np.float64).reshape(-1, 8, 3), pos


# ---------------------------------------------------------------------------------------
# projection + clip (reference: inline statements V3:565-569 and V3:584-592)
# ---------------------------------------------------------------------------------------
def project_points(points, TrVeloToRect, camera, depth_max=50.0, want_depth=True, device=0):
    """(u, v, depth, valid_indices) of f32[N,4] velodyne points: u, v int64 as in the reference
    (``np.round(...).astype(int)``), depth float64 (None unless want_depth), valid_indices =
    ``np.where(valid)[0]``."""
    ctx = get_context(device)
    ctx.set_camera(TrVeloToRect, camera.K, camera.width, camera.height, 0.0, float(depth_max))
    ctx.clear_masks()
    ctx.clear_boxes()
    r = ctx.run(_f32_points(points).reshape(-1, 4), want_float=want_depth, want_label=False)
    return r["u"].astype(np.int64), r["v"].astype(np.int64), (r["depth"] if want_depth else None), r["valid_idx"]


def per_car_depth_maps(points, TrVeloToRect, camera, masks, depth_max=30.0, device=0):
    """[(car_id, depthMap f64[H,W])] as seg_with_pointcloud.py:160-170 builds them (car_id = i + 1,
    ``depthMap[v,u] = depth`` of the last valid point in mask i at that pixel, 0 elsewhere).  The
    scatter -- a Python loop over every valid point per mask in the reference -- runs once on the
    GPU (the winner of a pixel does not depend on the mask); the per-mask select is elementwise."""
    ctx = get_context(device)
    ctx.set_camera(TrVeloToRect, camera.K, camera.width, camera.height, 0.0, float(depth_max))
    D, _ = ctx.depth_image(_f32_points(points).reshape(-1, 4))
    return [(i + 1, np.where(np.asarray(m) > 0.5, D, 0.0)) for i, m in enumerate(masks)]


# ---------------------------------------------------------------------------------------
# mask lookup (V3:211-233, cvs_erosion.py:148-162)
# ---------------------------------------------------------------------------------------
def extract_car_points_by_mask(points_valid, u_valid, v_valid, masks, camera, device=0):
    """One point array per mask: ``points_valid[mask[v_valid, u_valid] > 0.5]``, empty masks give
    a (0,3) float64 array.  The pixels are looked up by the same kernel the fused path uses
    (run on the already-projected pixels with an identity camera)."""
    pv = np.asarray(points_valid)
    n = pv.shape[0]
    ctx = get_context(device)
    ctx.set_camera(np.eye(4), np.eye(3), camera.width, camera.height, 0.0, 2.0)
    stack, _ = _mask_stack(masks, camera, resize_ctx=ctx)    # (V3:222: masks of another size are resized, on the GPU)
    M = stack.shape[0]
    sets = []
    if M == 0:
        return sets
    pix = np.zeros((n, 4), np.float32)
    pix[:, 0] = np.asarray(u_valid)
    pix[:, 1] = np.asarray(v_valid)
    pix[:, 2] = 1.0
    ctx.clear_boxes()
    for m0 in range(0, M, LPF_MAX_MASKS):
        ctx.set_masks(stack[m0:m0 + LPF_MAX_MASKS])
        r = ctx.run(pix, want_uv=False, want_label=False)
        for lst in r["inst_lists"]:
            sets.append(pv[lst] if len(lst) else np.array([]).reshape(0, 3))
    return sets


# ---------------------------------------------------------------------------------------
# box membership (V3:143-208)
# ---------------------------------------------------------------------------------------
def oriented_point_in_bbox(points, bbox_corners, device=0):
    """bool[k]: inside the three slabs c1-c0, c3-c0, c4-c0 (closed), as V3:167-204."""
    if len(points) == 0:
        return np.array([])
    p = _f32_points(points)
    return get_context(device).points_in_boxes(p, np.asarray(bbox_corners, np.float64)[None], oriented=True)[0]


def point_in_bbox(points, bbox_corners, device=0):
    """bool[k]: inside the axis-aligned hull of the 8 corners (closed), as V3:143-164."""
    if len(points) == 0:
        return np.array([])
    p = _f32_points(points)
    return get_context(device).points_in_boxes(p, np.asarray(bbox_corners, np.float64)[None], oriented=False)[0]


def _count_matrix(car_point_sets, corners, use_oriented, device=0, want_masks=False):
    """counts[car, box] (+ the per-point inside matrix and set offsets) with ONE kernel call."""
    sizes = [len(s) for s in car_point_sets]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    B = corners.shape[0]
    counts = np.zeros((len(sizes), B), np.int64)
    inside = None
    if off[-1] and B:
        cat = np.concatenate([_f32_points(s).reshape(-1, 3) for s in car_point_sets if len(s)], axis=0)
        inside = get_context(device).points_in_boxes(cat, corners, oriented=use_oriented)
        csum = np.concatenate([np.zeros((B, 1), np.int64), np.cumsum(inside, axis=1, dtype=np.int64)], axis=1)
        counts = (csum[:, off[1:]] - csum[:, off[:-1]]).T.copy()
    return counts, (inside if want_masks else None), off


def _best_box(counts_row):
    """First strict maximum starting from 0 (V3:353-376): (-1, 0) when no box holds a point."""
    best, idx = 0, -1
    for b, c in enumerate(counts_row):
        if c > best:
            best, idx = int(c), b
    return idx, best


def match_car_points_to_bboxes(car_point_sets, bboxes_3d, colors, min_points=10, use_oriented=True, device=0):
    """[(corners_velo, rgb_color, count)] for the matched cars (V3:236-290)."""
    matched = []
    if not bboxes_3d or len(car_point_sets) == 0:
        return matched
    corners, pos = _corners_velo(bboxes_3d)
    counts, _, _ = _count_matrix(car_point_sets, corners, use_oriented, device)
    for car_idx, car_points in enumerate(car_point_sets):
        if len(car_points) == 0:
            continue
        best, best_idx = 0, -1
        for j, c in enumerate(counts[car_idx]):
            if c > best and c >= min_points:
                best, best_idx = int(c), pos[j]
        if best_idx >= 0:
            col = colors[car_idx]
            matched.append((np.array(bboxes_3d[best_idx]["corners_velo"]), np.array([col[2], col[1], col[0]]) / 255.0, best))
            print(f"  Matched car {car_idx} to bbox {best_idx} with {best} points")
    return matched


def stats_from_counts(car_sizes, counts, colors, min_points=10, box_positions=None):
    """The reference's per-car dicts (cvs_erosion.py:165-229 key set) from integer counts:
    car_sizes[m] = len(car_point_sets[m]), counts[m, b] = points of car m inside box b."""
    out = []
    for car_idx, total in enumerate(car_sizes):
        total = int(total)
        if total == 0:
            continue
        j, best = _best_box(counts[car_idx]) if counts.shape[1] else (-1, 0)
        if j >= 0 and best >= min_points:
            inside, outside = best, total - best
            out.append({"car_id": car_idx, "matched_bbox_id": box_positions[j] if box_positions is not None else j,
                        "total_points": total, "points_inside_bbox": inside, "points_outside_bbox": outside,
                        "inside_percentage": (inside / total) * 100, "outside_percentage": (outside / total) * 100,
                        "color": colors[car_idx], "_best_col": j, "_best_count": best})
        else:
            out.append({"car_id": car_idx, "matched_bbox_id": -1, "total_points": total, "points_inside_bbox": 0,
                        "points_outside_bbox": total, "inside_percentage": 0.0, "outside_percentage": 100.0,
                        "color": colors[car_idx], "_best_col": -1, "_best_count": best})
    return out


def calculate_car_point_statistics(car_point_sets, bboxes_3d, colors, min_points=10, use_oriented=True,
                                   style="v3", device=0):
    """Per-car inside/outside statistics (V3:320-428; ``style='cvs'`` gives cvs_erosion.py:165-229's
    quieter variant without the array-valued keys).  Cars with no points get no row."""
    stats = []
    if not bboxes_3d or len(car_point_sets) == 0:
        return stats
    v3 = style == "v3"
    if v3:
        print(f"\n=== Car Point Statistics ===")
    print(f"Total car detections: {len(car_point_sets)}")
    print(f"Total 3D bounding boxes: {len(bboxes_3d)}")
    corners, pos = _corners_velo(bboxes_3d)
    counts, inside, off = _count_matrix(car_point_sets, corners, use_oriented, device, want_masks=v3)
    rows = stats_from_counts([len(s) for s in car_point_sets], counts, colors, min_points, pos)
    by_car = {r["car_id"]: r for r in rows}
    for car_idx, car_points in enumerate(car_point_sets):
        total = len(car_points)
        if total == 0:
            if v3:
                print(f"\nCar {car_idx}: No points detected")
            continue
        r = by_car[car_idx]
        j, best = r.pop("_best_col"), r.pop("_best_count")
        if v3:
            print(f"\nCar {car_idx}: {total} total points")
            if r["matched_bbox_id"] >= 0:
                r["corners_velo"] = np.array(bboxes_3d[r["matched_bbox_id"]]["corners_velo"])
                r["inside_mask"] = inside[j, off[car_idx]:off[car_idx + 1]]
                print(f"  ✓ Matched to 3D bbox {r['matched_bbox_id']}")
                print(f"  │ Points inside bbox:  {r['points_inside_bbox']:4d} ({r['inside_percentage']:5.1f}%)")
                print(f"  │ Points outside bbox: {r['points_outside_bbox']:4d} ({r['outside_percentage']:5.1f}%)")
                print(f"  └ Total points:        {total:4d} (100.0%)")
            else:
                r["corners_velo"] = None
                r["inside_mask"] = None
                print(f"  ✗ No matching 3D bbox found (best match: {best} points < {min_points} threshold)")
            r["car_points"] = car_points
        stats.append(r)
    return stats


# ---------------------------------------------------------------------------------------
# 2D IoU matching of V4 (V4:118-183) -- D x B scalars on the host
# ---------------------------------------------------------------------------------------
def calculate_iou_2d(box1, box2):
    x1a, y1a, x1b, y1b = box1
    x2a, y2a, x2b, y2b = box2
    xa, ya = max(x1a, x2a), max(y1a, y2a)
    xb, yb = min(x1b, x2b), min(y1b, y2b)
    if xb <= xa or yb <= ya:
        return 0.0
    inter = (xb - xa) * (yb - ya)
    union = (x1b - x1a) * (y1b - y1a) + (x2b - x2a) * (y2b - y2a) - inter
    return inter / union if union > 0 else 0.0


def match_detections_to_bboxes(boxes_2d, bboxes_3d, colors, camera, min_iou=0.25):
    """[(corners_velo, rgb_color)] per detection whose best projected box beats min_iou (V4:140-183)."""
    pairs = []
    if not bboxes_3d or len(boxes_2d) == 0:
        return pairs
    proj = []
    for bbox in bboxes_3d:
        if "corners_cam0" not in bbox:
            proj.append(None)
            continue
        if "_bbox2d" in bbox:                                 # projected on the GPU by prepare_boxes
            proj.append(bbox["_bbox2d"])
            continue
        u, v, depth = camera.cam2image(np.array(bbox["corners_cam0"]).T)
        front = depth > 0
        proj.append([np.min(u[front]), np.min(v[front]), np.max(u[front]), np.max(v[front])] if front.sum() else None)
    for det_idx, box in enumerate(boxes_2d):
        x1, y1, x2, y2 = box
        best, best_idx = 0, -1
        for j, pb in enumerate(proj):
            if pb is None:
                continue
            iou = calculate_iou_2d([x1, y1, x2, y2], pb)
            if iou > best and iou > min_iou:
                best, best_idx = iou, j
        if best_idx >= 0 and "corners_velo" in bboxes_3d[best_idx]:
            c = colors[det_idx]
            pairs.append((np.array(bboxes_3d[best_idx]["corners_velo"]), np.array([c[2], c[1], c[0]]) / 255.0))
    return pairs


def _projected_box_info(bbox, camera):
    """V5's project_3d_bbox_to_2d (V5:215-252) for one box dict: None when no corner is in front."""
    if "_bbox2d" in bbox:                                   # projected on the GPU by prepare_boxes
        bb = bbox["_bbox2d"]
    else:
        u, v, depth = camera.cam2image(np.array(bbox["corners_cam0"]).T)
        front = depth > 0
        bb = [np.min(u[front]), np.min(v[front]), np.max(u[front]), np.max(v[front])] if np.any(front) else None
    if bb is None:
        return None
    x0, y0, x1, y1 = bb
    return {"bbox": [x0, y0, x1, y1], "center": [(x0 + x1) / 2, (y0 + y1) / 2], "size": [x1 - x0, y1 - y0],
            "area": (x1 - x0) * (y1 - y0)}


def calculate_matching_score(detection_info, bbox_3d_info, weight_iou=0.5, weight_center=0.3, weight_size=0.2):
    """0.5 IoU + 0.3 centre proximity + 0.2 area ratio (V5:277-304)."""
    iou = calculate_iou_2d(detection_info["bbox"], bbox_3d_info["bbox"])
    dist = np.linalg.norm(np.array(detection_info["center"]) - np.array(bbox_3d_info["center"]))
    center_score = max(0, 1 - dist / 1000)
    det_area = detection_info["size"][0] * detection_info["size"][1]
    box_area = bbox_3d_info["area"]
    size_ratio = min(det_area, box_area) / max(det_area, box_area) if det_area > 0 and box_area > 0 else 0
    total = weight_iou * iou + weight_center * center_score + weight_size * size_ratio
    return total, {"iou": iou, "center_score": center_score, "size_score": size_ratio, "total_score": total}


def improved_match_detections_to_bboxes(boxes_2d, bboxes_3d, mask_colors, camera, min_score_threshold=0.3,
                                        min_iou_threshold=0.15):
    """V5's score matrix + Hungarian assignment (V5:307-416): matched boxes in the detection's colour,
    then every unmatched box with 'corners_velo' in light grey.  Host scalars (D x B), scipy's solver."""
    from scipy.optimize import linear_sum_assignment
    matched = []
    if not bboxes_3d or len(boxes_2d) == 0:
        print("[INFO] No detections or 3D bounding boxes to match")
        return matched
    print(f"[INFO] Matching {len(boxes_2d)} 2D detections with {len(bboxes_3d)} 3D bboxes")
    dets = []
    for box in boxes_2d:
        if len(box) == 4:
            x1, y1, x2, y2 = box
            dets.append({"bbox": [x1, y1, x2, y2], "center": [(x1 + x2) / 2, (y1 + y2) / 2], "size": [x2 - x1, y2 - y1],
                         "area": (x2 - x1) * (y2 - y1)})
    infos, valid_idx = [], []
    for j, bbox in enumerate(bboxes_3d):
        info = _projected_box_info(bbox, camera) if "corners_cam0" in bbox else None
        if info is not None:
            infos.append(info)
            valid_idx.append(j)
    if not infos:
        print("[WARN] No valid 3D bbox projections found")
        return matched
    cost = np.zeros((len(dets), len(infos)))
    details = {}
    for i, d in enumerate(dets):
        for j, b in enumerate(infos):
            score, det = calculate_matching_score(d, b)
            cost[i, j] = 1 - score
            details[(i, j)] = det
    rows, cols = linear_sum_assignment(cost)
    used = set()
    for i, j in zip(rows, cols):
        sc = details[(i, j)]
        if sc["total_score"] >= min_score_threshold and sc["iou"] >= min_iou_threshold:
            orig = valid_idx[j]
            used.add(orig)
            bbox = bboxes_3d[orig]
            if "corners_velo" in bbox:
                if i < len(mask_colors):
                    c = mask_colors[i]
                    color = np.array([c[2], c[1], c[0]], dtype=float) / 255.0
                else:
                    color = np.array([1.0, 0.0, 0.0])
                matched.append((np.array(bbox["corners_velo"]), color))
                print(f"[INFO] Matched detection {i} with 3D bbox {orig}")
                print(f"        Scores - IoU: {sc['iou']:.3f}, Center: {sc['center_score']:.3f}, "
                      f"Size: {sc['size_score']:.3f}, Total: {sc['total_score']:.3f}")
            else:
                print(f"[WARN] No Velodyne corners found for bbox {orig}")
        else:
            print(f"[INFO] Rejected match det{i}-bbox{j}: score={sc['total_score']:.3f}, IoU={sc['iou']:.3f}")
    for i, bbox in enumerate(bboxes_3d):
        if i not in used and "corners_velo" in bbox:
            matched.append((np.array(bbox["corners_velo"]), [0.7, 0.7, 0.7]))
            print(f"[INFO] Added unmatched 3D bbox {i} in default color")
    return matched


# ---------------------------------------------------------------------------------------
# box-view helpers of secondtest.py / V5 / firsttest.py (8 corners per box: host scalars)
# ---------------------------------------------------------------------------------------
_HSV_SECTORS = ((0, 3, 1), (2, 0, 1), (1, 0, 3), (1, 2, 0), (3, 1, 0), (0, 1, 2))   # (r, g, b) picks from (value, p, q, t)


def generate_consistent_colors(n_objects):
    """BGR 0-255 tuples, golden-angle hues (V5:88-121): hue = i*137.508 mod 360,
    saturation 0.8 + 0.1*(i%3), value 0.8 + 0.2*(i%2)."""
    out = []
    for i in range(n_objects):
        hue = (i * 137.508) % 360
        sat, val = 0.8 + (i % 3) * 0.1, 0.8 + (i % 2) * 0.2
        sector = int(hue / 60) % 6
        frac = (hue / 60) - sector
        comp = (val, val * (1 - sat), val * (1 - frac * sat), val * (1 - (1 - frac) * sat))
        r, g, b = (comp[k] for k in _HSV_SECTORS[sector])
        out.append((int(b * 255), int(g * 255), int(r * 255)))
    return out


def project_3d_bbox_to_2d(bbox_3d, camera, detailed=True):
    """Image-plane box of the corners with depth > 0.  detailed=True: V5:215-252 /
    secondtest.py:215-252 -> ({'bbox','center','size','area','avg_depth'}, corners_cam0);
    detailed=False: firsttest.py:172-193 -> ([x_min, y_min, x_max, y_max], corners_cam0).
    (None, None) when nothing is in front of the camera or the dict has no corners."""
    try:
        corners = np.array(bbox_3d["corners_cam0"])
        u, v, depth = camera.cam2image(corners.T)
        front = depth > 0
        if np.any(front):
            x0, x1 = np.min(u[front]), np.max(u[front])
            y0, y1 = np.min(v[front]), np.max(v[front])
            if not detailed:
                return [x0, y0, x1, y1], corners
            return {"bbox": [x0, y0, x1, y1], "center": [(x0 + x1) / 2, (y0 + y1) / 2], "size": [x1 - x0, y1 - y0],
                    "area": (x1 - x0) * (y1 - y0), "avg_depth": np.mean(depth[front])}, corners
    except Exception as e:
        print(f"[ERROR] Failed to project 3D bbox: {e}")
    return None, None


def is_bbox_in_camera_view(bbox_3d, camera, min_points_in_view=4, depth_range=(0.1, 100)):
    """(keep, info) of secondtest.py:277-359.  Corners count when depth lies in the closed
    depth_range; a box is dropped when none do ('all_behind_camera'), when fewer than
    min_points_in_view corners are inside the image AND the corners' pixel box misses the image
    ('no_intersection'), or when that pixel box is under 100 px^2 ('too_small')."""
    try:
        if "corners_cam0" not in bbox_3d:
            return False, {"reason": "no_corners"}
        u, v, depth = camera.cam2image(np.array(bbox_3d["corners_cam0"]).T)
        near = (depth >= depth_range[0]) & (depth <= depth_range[1])
        n_near = np.sum(near)
        if n_near == 0:
            return False, {"reason": "all_behind_camera", "depths": depth.tolist()}
        in_view = np.sum(near & (u >= 0) & (u < camera.width) & (v >= 0) & (v < camera.height))
        un, vn = u[near], v[near]
        if in_view < min_points_in_view:
            x0, x1, y0, y1 = np.min(un), np.max(un), np.min(vn), np.max(vn)
            if x1 < 0 or x0 >= camera.width or y1 < 0 or y0 >= camera.height:
                return False, {"reason": "no_intersection", "corners_in_view": in_view, "bbox_2d": [x0, y0, x1, y1]}
        if n_near >= 2:
            u_range, v_range = np.max(un) - np.min(un), np.max(vn) - np.min(vn)
            if u_range * v_range < 100:
                return False, {"reason": "too_small", "projected_area": u_range * v_range,
                               "u_range": u_range, "v_range": v_range}
        return True, {"reason": "valid", "corners_in_view": in_view, "corners_with_valid_depth": n_near,
                      "avg_depth": np.mean(depth[near]) if n_near > 0 else 0}
    except Exception as e:
        print(f"[ERROR] Error checking bbox visibility: {e}")
        return False, {"reason": "error", "error": str(e)}


def filter_bboxes_in_camera_view(bboxes_3d, camera, verbose=True):
    """(kept boxes in order, {'total','kept','filtered','filter_reasons'}) -- secondtest.py:362-419."""
    if not bboxes_3d:
        return [], {"total": 0, "kept": 0, "filtered": 0, "filter_reasons": {}}
    kept, reasons = [], {}
    for i, bbox in enumerate(bboxes_3d):
        ok, info = is_bbox_in_camera_view(bbox, camera)
        if ok:
            kept.append(bbox)
            if verbose:
                print(f"[INFO] Kept bbox {i}: {info['corners_in_view']} corners in view, "
                      f"avg depth: {info.get('avg_depth', 0):.2f}m")
            continue
        why = info["reason"]
        reasons[why] = reasons.get(why, 0) + 1
        if verbose:
            print(f"[INFO] Filtered bbox {i}: {why}")
            if why == "all_behind_camera" and info.get("depths"):
                print(f"        Depths: min={min(info['depths']):.2f}, max={max(info['depths']):.2f}")
            elif why == "no_intersection":
                print(f"        2D bbox: {info.get('bbox_2d', [])}")
            elif why == "too_small":
                print(f"        Projected area: {info.get('projected_area', 0):.1f} pixels")
    stats = {"total": len(bboxes_3d), "kept": len(kept), "filtered": len(bboxes_3d) - len(kept), "filter_reasons": reasons}
    if verbose:
        print(f"\n[STATS] BBox Filtering Results:")
        print(f"        Total: {stats['total']}")
        print(f"        Kept: {stats['kept']}")
        print(f"        Filtered: {stats['filtered']}")
        print(f"        Filter reasons: {stats['filter_reasons']}")
    return kept, stats


# ---------------------------------------------------------------------------------------
# exclusive labelling of Same_color.py:113-131 (first matching mask wins)
# ---------------------------------------------------------------------------------------
def _same_color_canvas(masks, camera):
    """Same_color.py:124 indexes every mask AT ITS OWN SIZE -- ``y < mask.shape[0] and x < mask.shape[1] and mask[y, x] > 0.5``, no
    resize -- so a mask that does not arrive at the camera's size counts in the pixels the two sizes share and nowhere else: the mask
    cropped / zero-padded to [H, W].  Masks at camera size (and GPU tensors at camera size) pass through."""
    H, W = camera.height, camera.width
    if _is_device_tensor(masks):
        import torch
        if masks.ndim != 3:
            raise ValueError("device masks must be [M,h,w]")
        if tuple(masks.shape[1:]) == (H, W):
            return _mask_stack(masks, camera)[0]
        src = masks if masks.dtype in (torch.float32, torch.uint8) else masks.to(torch.float32)
        canvas = torch.zeros((masks.shape[0], H, W), dtype=src.dtype, device=masks.device)
        h, w = min(H, masks.shape[1]), min(W, masks.shape[2])
        canvas[:, :h, :w] = src[:, :h, :w]
        return canvas
    planes = [np.asarray(mk) for mk in masks] if not isinstance(masks, np.ndarray) else list(masks)
    if not planes:
        return np.zeros((0, H, W), np.uint8)
    if all(pl.shape == (H, W) for pl in planes):
        return _mask_stack(np.stack(planes), camera)[0]
    flt = any(pl.dtype.kind == "f" for pl in planes)
    canvas = np.zeros((len(planes), H, W), np.float32 if flt else np.uint8)
    for i, pl in enumerate(planes):                          # (the reference's list may hold masks of different sizes)
        if pl.ndim != 2:
            raise ValueError("masks must be a list of [h,w] arrays or an [M,h,w] array")
        h, w = min(H, pl.shape[0]), min(W, pl.shape[1])
        canvas[i, :h, :w] = pl[:h, :w]
    return canvas


def label_points_first_match(points, TrVeloToRect, camera, masks, mask_colors=None, depth_max=30.0, device=0):
    """Same_color.py's per-point double loop as one fused GPU pass.  Returns a dict:
    ``car_idx`` (indices into points of valid points that lie in some mask, ascending),
    ``car_mask`` (the FIRST mask each of them matched, ``mask[y, x] > 0.5``), ``background_idx``
    (valid points in no mask), and ``colored_points`` / ``colored_colors`` / ``full_points`` as the
    reference accumulates them (colors = mask_colors[i] / 255.0 when mask_colors is given)."""
    p = _f32_points(points).reshape(-1, 4)
    m = _same_color_canvas(masks, camera)
    if m.shape[0] > LPF_MAX_MASKS:
        raise ValueError("at most %d masks per frame" % LPF_MAX_MASKS)
    ctx = get_context(device)
    ctx.set_camera(TrVeloToRect, camera.K, camera.width, camera.height, 0.0, float(depth_max))
    ctx.set_masks(m, binarize="gt0.5")
    ctx.clear_boxes()
    r = ctx.run(p, want_uv=False)
    ctx.clear_masks()
    vidx = r["valid_idx"]
    bits = r["label_bits"][vidx]
    hit = bits != 0
    low = bits[hit] & (~bits[hit] + np.uint32(1))                       # lowest set bit = first mask in list order
    first = np.log2(low.astype(np.float64)).astype(np.int64) if low.size else np.zeros(0, np.int64)
    out = {"car_idx": vidx[hit], "car_mask": first, "background_idx": vidx[~hit],
           "colored_points": p[vidx[hit], :3], "full_points": p[vidx[~hit], :3]}
    if mask_colors is not None:
        table = np.array([np.array(c) / 255.0 for c in mask_colors], np.float64).reshape(-1, 3)
        out["colored_colors"] = table[first]
    return out


# ---------------------------------------------------------------------------------------
# reporting (V3:431-468, cvs_erosion.py:232-295)
# ---------------------------------------------------------------------------------------
def print_summary_statistics(car_statistics):
    if not car_statistics:
        print("\nNo car statistics to display.")
        return
    print(f"\n{'=' * 60}")
    print(f"{'SUMMARY STATISTICS':^60}")
    print(f"{'=' * 60}")
    matched = [s for s in car_statistics if s["matched_bbox_id"] >= 0]
    print(f"Total cars detected: {len(car_statistics)}")
    print(f"Successfully matched: {len(matched)}")
    print(f"Unmatched: {len(car_statistics) - len(matched)}")
    if matched:
        print(f"\n{'Car ID':<8} {'BBox ID':<8} {'Total':<8} {'Inside':<8} {'Outside':<8} {'Inside %':<10}")
        print("-" * 60)
        for s in matched:
            print(f"{s['car_id']:<8} {s['matched_bbox_id']:<8} {s['total_points']:<8} {s['points_inside_bbox']:<8} "
                  f"{s['points_outside_bbox']:<8} {s['inside_percentage']:<10.1f}")
        tp = sum(s["total_points"] for s in matched)
        ti = sum(s["points_inside_bbox"] for s in matched)
        to = sum(s["points_outside_bbox"] for s in matched)
        avg = (ti / tp * 100) if tp > 0 else 0
        print("-" * 60)
        print(f"{'TOTAL':<8} {'':<8} {tp:<8} {ti:<8} {to:<8} {avg:<10.1f}")


CSV_COLUMNS = ("frame", "car_id", "matched_bbox_id", "total_points", "points_inside_bbox", "points_outside_bbox",
               "inside_percentage", "outside_percentage", "is_matched", "timestamp")


def csv_rows(car_statistics, frame_number, timestamp=None):
    """The rows append_to_master_csv writes (cvs_erosion.py:243-255)."""
    ts = timestamp if timestamp is not None else datetime.now().isoformat()
    return [{"frame": frame_number, "car_id": s["car_id"], "matched_bbox_id": s["matched_bbox_id"],
             "total_points": s["total_points"], "points_inside_bbox": s["points_inside_bbox"],
             "points_outside_bbox": s["points_outside_bbox"], "inside_percentage": round(s["inside_percentage"], 2),
             "outside_percentage": round(s["outside_percentage"], 2), "is_matched": s["matched_bbox_id"] >= 0,
             "timestamp": ts} for s in car_statistics]


def _csv_cell(v):
    """one value as pandas' to_csv writes it: bool -> True / False, integer -> digits, float -> shortest repr"""
    if isinstance(v, (bool, np.bool_)):
        return "True" if v else "False"
    if isinstance(v, (int, np.integer)):
        return int(v)
    if isinstance(v, (float, np.floating)):
        return repr(float(v))
    return v


def append_to_master_csv(car_statistics, frame_number, master_csv_path="results/master_car_statistics.csv", timestamp=None):
    if not car_statistics:
        return
    import csv
    d = os.path.dirname(master_csv_path)
    if d:
        os.makedirs(d, exist_ok=True)
    # The bytes pandas' DataFrame(rows).to_csv(index=False) writes (cvs_erosion.py:257-265) -- integers as they are, floats by their
    # shortest repr, booleans as True / False, minimal quoting, "\n" -- without building a DataFrame per frame: in a frame loop the
    # DataFrame cost more than the frame's kernels (tests/test_pipeline_host.py compares the two writers byte for byte).
    rows = csv_rows(car_statistics, frame_number, timestamp)
    new_file = not os.path.exists(master_csv_path)
    with open(master_csv_path, "a", newline="") as fh:
        w = csv.writer(fh, lineterminator="\n")
        if new_file:
            w.writerow(CSV_COLUMNS)
        for r in rows:
            w.writerow([_csv_cell(r[c]) for c in CSV_COLUMNS])
    if new_file:
        print(f"Created new master CSV: {master_csv_path}")
    else:
        print(f"Appended {len(rows)} rows to master CSV: {master_csv_path}")


def analyze_master_csv(master_csv_path="results/master_car_statistics.csv"):
    if not os.path.exists(master_csv_path):
        print(f"Master CSV file not found: {master_csv_path}")
        return
    import pandas as pd
    df = pd.read_csv(master_csv_path)
    print(f"\n{'=' * 60}")
    print(f"{'OVERALL ANALYSIS':^60}")
    print(f"{'=' * 60}")
    print(f"Total frames processed: {df['frame'].nunique()}")
    print(f"Total car detections: {len(df)}")
    print(f"Successfully matched cars: {df['is_matched'].sum()}")
    print(f"Unmatched cars: {(~df['is_matched']).sum()}")
    print(f"Average matching rate: {df['is_matched'].mean() * 100:.1f}%")
    m = df[df["is_matched"] == True]  # noqa: E712  (same comparison as the reference)
    if len(m) > 0:
        print(f"\nMatched Cars Statistics:")
        print(f"Average points per car: {m['total_points'].mean():.1f}")
        print(f"Average inside percentage: {m['inside_percentage'].mean():.1f}%")
        print(f"Min inside percentage: {m['inside_percentage'].min():.1f}%")
        print(f"Max inside percentage: {m['inside_percentage'].max():.1f}%")
    return df


# ---------------------------------------------------------------------------------------
# the fused per-frame path: one batched launch set for many frames
# ---------------------------------------------------------------------------------------
def default_colors(n):
    """(int(i*60)%255, int(i*120)%255, int(i*180)%255), V3:100."""
    return [(int(i * 60) % 255, int(i * 120) % 255, int(i * 180) % 255) for i in range(n)]


class _LiveScanPoints:
    """The pinned points of a reader's Scan for the lazy gathers: indexable like the array while the scan is the reader's current one,
    an LpfError afterwards (its buffers hold another scan by then)."""
    __slots__ = ("scan",)

    def __init__(self, scan):
        self.scan = scan

    def __getitem__(self, key):
        self.scan._check_live()
        return self.scan.points[key]


class FrameResult(dict):
    """run_frames' dict of one frame.  The integers the kernels produced (valid_indices, count_mb, car_statistics, n_valid) are there
    at once; the arrays of the reference's types that are GATHERS or CASTS of them -- ``points_valid`` (V3:592), ``car_point_sets``
    (V3:228), ``u_valid`` / ``v_valid`` as int64 (V3:590-591), ``bg_assigned`` (V4:290-298) -- are made when they are first read (and
    kept): a caller that writes the CSV never pays for 25 000-row fancy-index gathers it does not look at.  Every way of looking at a
    dict sees the same keys and values as before."""
    __slots__ = ("_lazy",)

    def __init__(self, eager, lazy):
        super().__init__(eager)
        self._lazy = dict(lazy)                              # key -> zero-argument callable

    def __missing__(self, key):
        make = self._lazy.pop(key)                           # KeyError for a key that is neither
        val = make()
        self[key] = val
        return val

    def _all(self):
        for k in list(self._lazy):
            self[k]                                          # noqa: B018  (materialises)
        return self

    def get(self, key, default=None):
        return self[key] if (key in self._lazy or dict.__contains__(self, key)) else default

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self._lazy

    def __iter__(self):
        return dict.__iter__(self._all())

    def __len__(self):
        return dict.__len__(self) + len(self._lazy)

    def keys(self):
        return dict.keys(self._all())

    def values(self):
        return dict.values(self._all())

    def items(self):
        return dict.items(self._all())

    def copy(self):
        return dict(self._all())

    def __eq__(self, other):
        return dict.__eq__(self._all(), other)

    __hash__ = None

    def __repr__(self):
        return dict.__repr__(self._all())

    def __reduce__(self):                                    # pickles (and deep-copies) as the plain dict it stands for
        return (dict, (dict(self._all()),))


class _DevicePoints:
    """points[idx, :3] of a float32 [N,4] torch tensor on the GPU, as a NumPy array: the gather runs where the points are"""

    def __init__(self, t):
        self.t = t

    def __getitem__(self, key):
        idx, cols = key
        import torch
        i = torch.from_numpy(np.ascontiguousarray(idx, dtype=np.int64)).to(self.t.device)
        return self.t[i][:, cols].cpu().numpy()


class FrameInputs:
    """What one frame hands to the hot path: velodyne points, detection masks and the
    visible boxes already in velodyne coordinates."""

    def __init__(self, frame, points, masks=None, bboxes_3d=None, colors=None, boxes_2d=None):
        self.frame = frame
        # (a Scan of the read-ahead reader or a float32 [N,4] torch tensor already on the GPU is taken as it is: no host copy)
        self.points = points if (isinstance(points, Scan) or _is_device_tensor(points)) else _f32_points(points).reshape(-1, 4)
        self.masks = masks
        self.bboxes_3d = bboxes_3d if bboxes_3d is not None else []
        n = 0 if masks is None else len(masks)
        self.colors = colors if colors is not None else default_colors(n)
        self.boxes_2d = boxes_2d


def _host_masks_to_device_batch(stacks, M, H, W, ctx):
    """[F,M,H,W] torch tensor on the context's GPU holding the frames' host masks (float32 if any frame's are, else uint8; frames
    with fewer detections padded with empty masks), or None when torch / its GPU support is not there."""
    try:
        import torch
        if not torch.cuda.is_available():
            return None
    except Exception:
        return None
    flt = any(s.dtype == np.float32 for s in stacks if s.shape[0])
    dev = torch.device("cuda", ctx.device)
    ragged = any(s.shape[0] != M for s in stacks)
    t = (torch.zeros if ragged else torch.empty)((len(stacks), M, H, W), dtype=torch.float32 if flt else torch.uint8, device=dev)
    for i, s in enumerate(stacks):
        if s.shape[0]:
            t[i, :s.shape[0]].copy_(torch.from_numpy(s if s.dtype == (np.float32 if flt else np.uint8) else s.astype(np.float32 if flt else np.uint8)))
    ctx.wait_for_stream(torch.cuda.current_stream(dev).cuda_stream)    # the copies were queued on torch's stream
    return t


def run_frames(frames, TrVeloToRect, camera, depth_max=50.0, min_points=10, use_oriented=True,
               erode_iters=0, v3_pipeline=False, device=0, ctx=None, gather_scans=True):
    """Projection + clip + mask lookup + box counting + best-box scan for a list of
    FrameInputs in ONE batched call (frames are independent units).  Returns one dict per
    frame: valid_indices, u_valid, v_valid, points_valid, car_point_sets, bg_assigned,
    count_mb, car_statistics (cvs_erosion key set) -- the integers are the kernels' output,
    the dicts are assembled here.  Frames whose points are a read-ahead reader's ``Scan`` have their gathers (``points_valid``,
    ``car_point_sets``) made before the call returns, because the reader recycles the scan's buffers when it moves on;
    ``gather_scans=False`` leaves them lazy like everyone else's -- reading them after the reader has moved on raises."""
    if not frames:
        return []
    ctx = ctx or get_context(device)
    H, W = camera.height, camera.width
    ctx.set_camera(TrVeloToRect, camera.K, W, H, 0.0, float(depth_max))
    # (masks of another size than the camera's: cv2.resize as V3:222, on the GPU; with the V3 erosion block in force such masks are
    #  eroded at their own size first, as V3:82-97 does before V3:222 -- and then so are the batch's other frames, by the same chain)
    def _off_size(mk):
        shp = tuple(getattr(mk, "shape", ())) if mk is not None else ()
        return len(shp) == 3 and shp[0] > 0 and shp[1:] != (H, W)
    chain_all = bool(erode_iters or v3_pipeline) and any(_off_size(f.masks) for f in frames)
    stacks = [_mask_stack(f.masks if f.masks is not None else [], camera, resize_ctx=ctx, erode_iters=erode_iters, v3_pipeline=v3_pipeline,
                          force_chain=chain_all)[0] for f in frames]
    if chain_all:                                                        # the erosion has been done: the masks below are uint8 0 / 1 at camera size
        erode_iters, v3_pipeline = 0, False
    M = max(s.shape[0] for s in stacks)
    if M > LPF_MAX_MASKS:
        # The reference loops over every mask (V3:220), with no bound; a launch labels a point with one bit per mask in a
        # 32-bit word.  More detections than that: the frames run once per group of 32 masks (same points, same boxes) and the
        # per-detection results are put together -- everything per point (pixels, valid indices) is the same in every pass.
        return _run_frames_in_mask_groups(frames, stacks, TrVeloToRect, camera, depth_max, min_points, use_oriented, erode_iters,
                                          v3_pipeline, device, ctx)
    on_gpu = [_is_device_tensor(s) for s in stacks]
    if any(on_gpu):                                                      # YOLO's masks still on the GPU: no host round trip
        import torch
        if not all(on_gpu) or any(s.shape[0] != M or s.dtype != stacks[0].dtype for s in stacks):
            raise NotImplementedError("device masks: every frame of a batch needs the same detection count and dtype")
        batch = stacks[0][None] if len(stacks) == 1 else torch.stack(stacks)
        ctx.wait_for_stream(torch.cuda.current_stream(batch.device).cuda_stream)    # the masks were produced on torch's stream
        stacks = [np.empty((M, 0, 0), np.uint8)] * len(stacks)          # only their detection count is used below
    elif len(stacks) > 1 and (batch := _host_masks_to_device_batch(stacks, M, H, W, ctx)) is not None:
        # several frames of host masks: each frame's masks go to their place in ONE device tensor -- no np.stack of the batch on the
        # host first (32 frames of five float masks are 340 MB: the copy cost more than everything else in the call)
        stacks = [np.empty((s.shape[0], 0, 0), np.uint8) for s in stacks]
    elif all(s.shape[0] == M and s.dtype == (np.float32 if any(t.dtype == np.float32 for t in stacks if t.shape[0]) else np.uint8) for s in stacks):
        batch = stacks[0][None] if len(stacks) == 1 else np.stack(stacks)   # the usual case: no padding, no extra copy
    else:                                                               # ragged detection counts: pad with empty masks
        dt = np.float32 if any(t.dtype == np.float32 for t in stacks if t.shape[0]) else np.uint8
        batch = np.zeros((len(frames), M, H, W), dt)
        for i, s in enumerate(stacks):
            if s.shape[0]:
                batch[i, :s.shape[0]] = s
    corners, positions = [], []
    for f in frames:
        c, pos = _corners_velo(f.bboxes_3d)
        corners.append(c)
        positions.append(pos)
    ctx.set_masks(batch, erode_iters=erode_iters, v3_pipeline=v3_pipeline, lend=True)   # (the run follows in this call: GPU masks can be lent)
    ctx.set_boxes(corners, oriented=use_oriented)
    # only the valid points' pixels and labels are used below: fetch those (a quarter of the dense arrays on real frames)
    # (results arrive in page-locked buffers the context reuses: what is handed to the caller is copied out of them below)
    res = ctx.run_batch([f.points for f in frames], want_uv=False, want_label=False, want_valid_uv=True, pinned=True)
    out = []
    for f, r, s, pos in zip(frames, res, stacks, positions):
        m = s.shape[0]
        # (the result arrays live in page-locked buffers the context reuses: what outlives this call is copied out of them here -- the
        #  compact lists, a few hundred KB -- and the gathers / casts of the reference's types are made from those copies when read)
        vi = r["valid_idx"].copy()
        uvv = r["uv_valid"].copy()                                       # int32 [n_valid, 2]: one contiguous copy
        labv = r["label_valid"].copy()
        lists = [l.copy() for l in r["inst_lists"][:m]]
        is_scan = isinstance(f.points, Scan)
        host_pts = _LiveScanPoints(f.points) if is_scan else f.points    # Scan: pinned copy of the file, while the reader has not moved on
        if _is_device_tensor(host_pts):                                  # points that live on the GPU: the gathers run there when asked for
            host_pts = _DevicePoints(host_pts)
        stats = []
        if f.bboxes_3d and m:
            stats = stats_from_counts(r["inst_count"][:m], r["count_mb"][:m], f.colors, min_points, pos)
            for d in stats:
                d.pop("_best_col"), d.pop("_best_count")
        lazy = dict(u_valid=lambda uvv=uvv: uvv[:, 0].astype(np.int64), v_valid=lambda uvv=uvv: uvv[:, 1].astype(np.int64),
                    points_valid=lambda p=host_pts, vi=vi: p[vi, :3],
                    car_point_sets=lambda p=host_pts, ls=lists: [p[l, :3] if len(l) else np.array([]).reshape(0, 3) for l in ls],
                    bg_assigned=lambda labv=labv: labv != 0)
        fr = FrameResult(dict(frame=f.frame, valid_indices=vi, count_mb=r["count_mb"][:m].copy(), car_statistics=stats, n_valid=r["n_valid"]), lazy)
        if is_scan and gather_scans:
            fr._all()                                        # (a Scan's pinned points are recycled when the reader moves on: gather now)
        out.append(fr)
    return out


def _run_frames_in_mask_groups(frames, stacks, TrVeloToRect, camera, depth_max, min_points, use_oriented, erode_iters, v3_pipeline,
                               device, ctx):
    """run_frames for frames with more than LPF_MAX_MASKS detections: one pass per group of 32 masks, results merged."""
    M = max(s.shape[0] for s in stacks)
    merged = None
    for g0 in range(0, M, LPF_MAX_MASKS):
        part = [FrameInputs(f.frame, f.points, s[g0:g0 + LPF_MAX_MASKS], f.bboxes_3d, f.colors[g0:g0 + LPF_MAX_MASKS], f.boxes_2d)
                for f, s in zip(frames, stacks)]
        res = run_frames(part, TrVeloToRect, camera, depth_max, min_points, use_oriented, erode_iters, v3_pipeline, device, ctx)
        if merged is None:
            merged = res
            continue
        for acc, r in zip(merged, res):
            acc["car_point_sets"] += r["car_point_sets"]
            acc["bg_assigned"] = acc["bg_assigned"] | r["bg_assigned"]
            acc["count_mb"] = np.concatenate([acc["count_mb"], r["count_mb"]], axis=0)
            for d in r["car_statistics"]:
                d["car_id"] += g0                            # car ids count the detections of the whole frame (V3:330)
            acc["car_statistics"] += r["car_statistics"]
    return merged


def stream_frames(scan_paths, inputs_for, TrVeloToRect, camera, depth_max=50.0, min_points=10, use_oriented=True,
                  erode_iters=0, v3_pipeline=False, device=0, n_buffers=3, max_points=None, box_paths=None, announce=None, gather=True):
    """The frame loop with read-ahead: scans are read and moved to HBM by the native reader
    (lpf_reader_*) while earlier frames are processed; yields run_frames' dict per frame.
    ``inputs_for(i, path)`` returns ``(frame_id, masks, bboxes_3d, colors)`` or None to skip the
    frame (the reference's ``continue`` rules); it runs while the scan is still being fetched.
    A missing scan prints the reference's message (cvs_erosion.py:326-330) and is skipped.
    With ``box_paths`` (one ``BBoxes_<frame>.json`` per scan) the reader's worker parses the box file beside the scan
    (lpf_reader_submit_frame) and ``inputs_for(i, path, scan)`` is called once the scan is there, ``scan.boxes_state`` /
    ``scan.box_index`` / ``scan.boxes_cam0`` holding the result; the scans and box files of the NEXT frames are being fetched
    meanwhile.  ``announce(i, path)`` runs before the scan is waited for (the reference's "Processing frame" line).
    ``gather=False``: the consumer only reads the statistics (process_frames writes the CSV): ``points_valid`` / ``car_point_sets``
    are not gathered from the scan before the reader recycles it (run_frames' ``gather_scans``)."""
    scan_paths = [os.fspath(p) for p in scan_paths]
    if max_points is None:
        sizes = [os.path.getsize(p) // 16 for p in scan_paths if os.path.isfile(p)]
        max_points = max(sizes + [1])
    ctx = get_context(device)
    with ScanReader(ctx, scan_paths, n_buffers=n_buffers, max_points=max_points, box_paths=box_paths) as reader:
        for i, path in enumerate(scan_paths):
            if announce is not None:
                announce(i, path)
            inputs = inputs_for(i, path) if box_paths is None else None
            try:
                scan = next(reader)
            except RuntimeError as e:
                print(f"Failed to load frame {os.path.basename(path)}: {e}")
                continue
            if box_paths is not None:
                inputs = inputs_for(i, path, scan)
            if inputs is None:
                continue
            frame_id, masks, boxes, colors = inputs
            yield run_frames([FrameInputs(frame_id, scan, masks, boxes, colors)], TrVeloToRect, camera, depth_max,
                             min_points, use_oriented, erode_iters, v3_pipeline, device, ctx, gather_scans=gather)[0]


# ---------------------------------------------------------------------------------------
# entry points (frame loops of cvs_erosion.py:298-379 and V3:516-641)
# ---------------------------------------------------------------------------------------
def sequence_setup(kitti360_path, seq=0, cam_id=0):
    """camera, TrVeloToCam, TrVeloToRect and the velodyne reader, composed as V3:520-535."""
    sequence = "2013_05_28_drive_%04d_sync" % seq
    camera = kitti360.CameraPerspective(kitti360_path, sequence, cam_id)
    velo_to_cam, velo_to_rect = kitti360.velo_to_rect_transforms(kitti360_path, camera, cam_id)
    velo = kitti360.Kitti360Viewer3DRaw(seq=seq, root_dir=kitti360_path)
    return sequence, camera, velo_to_cam, velo_to_rect, velo


def _boxes_of_file(json_path, camera, velo_to_cam, keep_all=False, parsed=None):
    """load_bounding_boxes (V3:31-38) + prepare_boxes for callers that keep the boxes to themselves; None = the reference's
    ``if not bboxes_3d_raw: continue`` (cvs_erosion.py:334-335: no file -- its message is printed -- or an empty list).  The file is
    parsed by the library (``parsed`` = the read-ahead reader's result for it, else lpf_parse_boxes_json now); a file that is not the
    plain schema goes through json.load as before."""
    state, index, corners = parsed if parsed is not None else _native.parse_boxes_file(json_path)
    if state == _native.BOXES_PARSED:
        return prepare_boxes_from_arrays(index, corners, camera, velo_to_cam, keep_all=keep_all) if len(index) else None
    raw = kitti360.load_bounding_boxes(json_path)           # absent: prints the reference's message, []; other schema: json.load
    if not raw:
        return None
    return prepare_boxes(raw, camera, velo_to_cam, keep_all=keep_all, as_arrays=True)


def collect_frame_inputs(kitti360_path, seq, cam_id, segmenter, image_loader, camera, velo_to_cam, velo, frames=None,
                         keep_all_boxes=False, boxes_as_arrays=False):
    """iter_frame_inputs as a list (every frame's scan and masks in memory at once: for a handful of frames)."""
    return list(iter_frame_inputs(kitti360_path, seq, cam_id, segmenter, image_loader, camera, velo_to_cam, velo, frames,
                                  keep_all_boxes, boxes_as_arrays))


def _batches(items, n):
    """Lists of up to n consecutive items of an iterable, made as they are asked for."""
    batch = []
    for it in items:
        batch.append(it)
        if len(batch) >= n:
            yield batch
            batch = []
    if batch:
        yield batch


def iter_frame_inputs(kitti360_path, seq, cam_id, segmenter, image_loader, camera, velo_to_cam, velo, frames=None,
                      keep_all_boxes=False, boxes_as_arrays=False):
    """The reference's per-frame loading + skip rules (cvs_erosion.py:320-369), one FrameInputs at a time: a frame is
    dropped when its scan, its box file, its image or its detections are missing.  (A generator: a frame loop over a whole drive
    holds one batch of scans and masks, not the drive.)"""
    sequence = "2013_05_28_drive_%04d_sync" % seq
    bbox_dir = os.path.join(kitti360_path, "bboxes_3D_cam0")
    todo = velo.available_frames() if frames is None else list(frames)
    print(f"Found {len(todo)} frames to process")
    for frame in todo:
        print(f"\nProcessing frame {frame}...")
        try:
            points = velo.loadVelodyneData(frame)
        except Exception as e:  # same breadth as the reference
            print(f"Failed to load frame {frame}: {e}")
            continue
        if boxes_as_arrays:                                 # (the caller keeps the boxes to itself: the library parses the file)
            boxes = _boxes_of_file(os.path.join(bbox_dir, f"BBoxes_{frame}.json"), camera, velo_to_cam, keep_all_boxes)
            if boxes is None:
                continue
        else:
            raw = kitti360.load_bounding_boxes(os.path.join(bbox_dir, f"BBoxes_{frame}.json"))
            if not raw:
                continue
            boxes = prepare_boxes(raw, camera, velo_to_cam, keep_all=keep_all_boxes)
        image_path = os.path.join(kitti360_path, "data_2d_raw", sequence, f"image_{cam_id:02d}",
                                  "data_rect" if cam_id in [0, 1] else "data_rgb", f"{frame:010d}.png")
        if not os.path.isfile(image_path):
            continue
        seg = segmenter(image_loader(image_path) if image_loader else image_path)
        _, masks, colors, boxes_2d, _ = seg
        if masks is None or len(masks) == 0:
            continue
        yield FrameInputs(frame, points, masks, boxes, colors, boxes_2d)


def process_frames(seq=0, cam_id=0, segmenter=None, image_loader=None, kitti360_path=None,
                   master_csv_path="results/master_car_statistics.csv", frames=None, batch_frames=32,
                   erode_iters=0, v3_pipeline=False, device=0, timestamp=None, read_ahead=True):
    """cvs_erosion.process_frames (cvs_erosion.py:298-379): writes the master CSV and prints the
    overall analysis.  ``segmenter(image) -> (img, masks, colors, boxes, confidences)`` is the
    YOLO stage (unchanged subsystem); pass masks it already eroded, or raw masks plus
    ``erode_iters=1, v3_pipeline=True`` to erode on the GPU.  ``read_ahead=True`` (the default) processes frame
    by frame, as the reference's loop does -- each frame's rows are appended before the next frame is looked at -- with the native
    reader fetching the next scans and parsing the next box files meanwhile; ``read_ahead=False`` reads ``batch_frames`` frames with
    NumPy and runs them as one launch (same CSV; its lines are printed batch by batch)."""
    if segmenter is None:
        raise ValueError("process_frames needs the segmentation callable (YOLO stays outside this package)")
    root = kitti360_path or os.environ["KITTI360_DATASET"]
    sequence, camera, velo_to_cam, velo_to_rect, velo = sequence_setup(root, seq, cam_id)
    if read_ahead:
        todo = velo.available_frames() if frames is None else list(frames)
        print(f"Found {len(todo)} frames to process")

        box_paths = [os.path.join(root, "bboxes_3D_cam0", f"BBoxes_{f}.json") for f in todo]

        def inputs_for(i, path, scan):
            frame = todo[i]
            # (the box file was parsed by the reader's worker while earlier frames ran: json.load of it was most of a frame's host time)
            boxes = _boxes_of_file(box_paths[i], camera, velo_to_cam, parsed=(scan.boxes_state, scan.box_index, scan.boxes_cam0))
            if boxes is None:
                return None
            image_path = os.path.join(root, "data_2d_raw", sequence, f"image_{cam_id:02d}",
                                      "data_rect" if cam_id in [0, 1] else "data_rgb", f"{frame:010d}.png")
            if not os.path.isfile(image_path):
                return None
            _, masks, colors, _, _ = segmenter(image_loader(image_path) if image_loader else image_path)
            if masks is None or len(masks) == 0:
                return None
            return frame, masks, boxes, colors

        paths = [os.path.join(velo.raw3DPcdPath, "%010d.bin" % f) for f in todo]
        for r in stream_frames(paths, inputs_for, velo_to_rect, camera, 50.0, 10, True, erode_iters, v3_pipeline, device,
                               box_paths=box_paths, announce=lambda i, path: print(f"\nProcessing frame {todo[i]}..."), gather=False):
            if r["n_valid"] and r["car_statistics"]:
                append_to_master_csv(r["car_statistics"], r["frame"], master_csv_path, timestamp)
        return analyze_master_csv(master_csv_path)
    items = iter_frame_inputs(root, seq, cam_id, segmenter, image_loader, camera, velo_to_cam, velo, frames, boxes_as_arrays=True)
    for batch in _batches(items, batch_frames):              # (one batch of scans and masks in memory at a time)
        for r in run_frames(batch, velo_to_rect, camera, 50.0, 10, True, erode_iters, v3_pipeline, device):
            if r["n_valid"] == 0:
                continue
            if r["car_statistics"]:
                append_to_master_csv(r["car_statistics"], r["frame"], master_csv_path, timestamp)
    return analyze_master_csv(master_csv_path)


def process_frame_with_statistics(seq=0, cam_id=0, segmenter=None, image_loader=None, kitti360_path=None,
                                  visualizer=None, frames=None, erode_iters=0, v3_pipeline=False, device=0):
    """V3's entry point (V3:516-641) without the blocking Open3D window: per frame it prints the
    statistics table and hands (frame, car_statistics, points_valid, bg_assigned) to
    ``visualizer`` when one is given.  bg_assigned is V4's vectorised form of V3:609-616."""
    if segmenter is None:
        raise ValueError("process_frame_with_statistics needs the segmentation callable")
    root = kitti360_path or os.environ["KITTI360_DATASET"]
    _, camera, velo_to_cam, velo_to_rect, velo = sequence_setup(root, seq, cam_id)
    items = collect_frame_inputs(root, seq, cam_id, segmenter, image_loader, camera, velo_to_cam, velo, frames)
    results = []
    for r, item in zip(run_frames(items, velo_to_rect, camera, 50.0, 10, True, erode_iters, v3_pipeline, device), items):
        if r["n_valid"] == 0:
            continue
        print_summary_statistics(r["car_statistics"])
        if visualizer is not None:
            visualizer(r["frame"], r["car_statistics"], r["points_valid"], r["bg_assigned"])
        results.append(r)
    return results


def process_frame(seq=0, cam_id=0, segmenter=None, image_loader=None, kitti360_path=None, visualizer=None,
                  frames=None, device=0):
    """V4's entry point (V4:213-336): depth < 30 clip, per-mask point sets, ``bg_assigned`` and the
    2D-IoU box matching; the Open3D window is replaced by the optional ``visualizer`` callable,
    which receives (frame, car_point_sets, colors, remaining_points, matched_pairs)."""
    if segmenter is None:
        raise ValueError("process_frame needs the segmentation callable")
    root = kitti360_path or os.environ["KITTI360_DATASET"]
    _, camera, velo_to_cam, velo_to_rect, velo = sequence_setup(root, seq, cam_id)
    items = collect_frame_inputs(root, seq, cam_id, segmenter, image_loader, camera, velo_to_cam, velo, frames)
    results = []
    for r, item in zip(run_frames(items, velo_to_rect, camera, 30.0, 10, True, 0, False, device), items):
        if r["n_valid"] == 0:
            continue
        r["remaining_points"] = r["points_valid"][~r["bg_assigned"]]
        r["matched_pairs"] = match_detections_to_bboxes(item.boxes_2d, item.bboxes_3d, item.colors, camera)
        print(f"Visualizing frame {r['frame']} with {sum(len(s) > 0 for s in r['car_point_sets']) + 1 + len(r['matched_pairs'])} objects")
        if visualizer is not None:
            visualizer(r["frame"], r["car_point_sets"], item.colors, r["remaining_points"], r["matched_pairs"])
        results.append(r)
    return results


def projectVeloToImage(seq=0, cam_id=0, segmenter=None, image_loader=None, kitti360_path=None, visualizer=None,
                       frames=None, device=0):
    """V5's entry point (V5:419-571): every annotated box (no visibility filter), depth < 30 clip,
    per-mask point sets, ``bg_assigned`` and the score + Hungarian box matching; ``visualizer`` receives
    (frame, car_point_sets, colors, remaining_points, matched_pairs) instead of the Open3D window."""
    if segmenter is None:
        raise ValueError("projectVeloToImage needs the segmentation callable")
    root = kitti360_path or os.environ["KITTI360_DATASET"]
    _, camera, velo_to_cam, velo_to_rect, velo = sequence_setup(root, seq, cam_id)
    items = collect_frame_inputs(root, seq, cam_id, segmenter, image_loader, camera, velo_to_cam, velo, frames,
                                 keep_all_boxes=True)
    results = []
    for r, item in zip(run_frames(items, velo_to_rect, camera, 30.0, 10, True, 0, False, device), items):
        print(f"[DEBUG] Frame {r['frame']}: {r['n_valid']} points passed validation filter")
        if r["n_valid"] == 0:
            print(f"[WARN] No valid LiDAR points in frame {r['frame']}")
            continue
        r["remaining_points"] = r["points_valid"][~r["bg_assigned"]]
        r["matched_pairs"] = improved_match_detections_to_bboxes(item.boxes_2d, item.bboxes_3d, item.colors, camera)
        if visualizer is not None:
            visualizer(r["frame"], r["car_point_sets"], item.colors, r["remaining_points"], r["matched_pairs"])
        results.append(r)
    return results
