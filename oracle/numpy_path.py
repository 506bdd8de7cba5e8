"""The reference's NumPy statements for K1-K7, as one function (TEST / BASELINE ONLY).

This is what the reference executes per frame on the host (V3:565-569, 584-592,
211-233, 167-204, 344-379 -- paths relative to /root/reference/Coding_testes),
restated with the same array expressions and dtypes so that its wall time is the
"reference NumPy path" of BASELINE.md (B0).  cv2.resize is the identity at equal size
(masks are H x W), so it is omitted.  Only tests/ and bench.py's cpu_baseline leg
import this module.
"""
import numpy as np


def cam2image(K3, points):
    """kitti360scripts CameraPerspective.cam2image on f64 [3,N]."""
    proj = np.matmul(K3.reshape(1, 3, 3), points[None])
    depth = proj[:, 2, :]
    depth[depth == 0] = -1e-6
    with np.errstate(invalid="ignore", over="ignore", divide="ignore"):
        u = np.round(proj[:, 0, :] / np.abs(depth)).astype(int)
        v = np.round(proj[:, 1, :] / np.abs(depth)).astype(int)
    return u[0], v[0], depth[0]


def oriented_point_in_bbox(points, c):
    if len(points) == 0:
        return np.array([])
    v1, v2, v3 = c[1] - c[0], c[3] - c[0], c[4] - c[0]
    rel = points - c[0]
    p1 = np.dot(rel, v1) / np.dot(v1, v1)
    p2 = np.dot(rel, v2) / np.dot(v2, v2)
    p3 = np.dot(rel, v3) / np.dot(v3, v3)
    return (p1 >= 0) & (p1 <= 1) & (p2 >= 0) & (p2 <= 1) & (p3 >= 0) & (p3 <= 1)


def frame_path(points, T, K3, W, H, dmax, masks, corners_velo, min_points=10):
    """points f32[N,4]; masks [M,H,W] (uint8 or float); corners_velo f64[B,8,3].
    Returns (u, v, valid_indices, mask_index_lists, count_mb, best_box, best_cnt)."""
    points_homo = points.copy()
    points_homo[:, 3] = 1
    pointsCam = np.matmul(T, points_homo.T).T[:, :3]
    u, v, depth = cam2image(K3, pointsCam.T)
    valid = (u >= 0) & (u < W) & (v >= 0) & (v < H) & (depth > 0) & (depth < dmax)
    valid_indices = np.where(valid)[0]
    u_valid, v_valid = u[valid], v[valid]
    points_valid = points[valid_indices, :3]
    lists, sets = [], []
    for mask in masks:
        sel = mask.astype(np.uint8)[v_valid, u_valid] > 0.5
        lists.append(valid_indices[sel])
        sets.append(points_valid[sel] if np.count_nonzero(sel) > 0 else np.array([]).reshape(0, 3))
    M, B = len(masks), len(corners_velo)
    count = np.zeros((M, B), np.int64)
    best_box = np.full(M, -1, np.int32)
    best_cnt = np.zeros(M, np.int64)
    for m, car_points in enumerate(sets):
        if len(car_points) == 0:
            continue
        for b in range(B):
            inside = oriented_point_in_bbox(car_points, corners_velo[b])
            n = int(np.sum(inside))
            count[m, b] = n
            if n > best_cnt[m]:
                best_cnt[m], best_box[m] = n, b
    return u, v, valid_indices, lists, count, best_box, best_cnt


def prepare_boxes(corners_cam0, K3, W, H, TrVeloToCam):
    """The reference's per-frame box preparation, restated: filter_visible_bboxes (V3:121-140 -- the cam-0 corners projected
    with cam2image WITHOUT R_rect, kept when at least 2 corners have depth > 0.1 inside the image) and
    transform_bboxes_to_velodyne (V3:41-52 -- inv(TrVeloToCam) . [c 1]).  corners_cam0: f64 [B,8,3].
    Returns (visible bool[B], corners_velo f64[B,8,3] of ALL boxes).  Pinned against the reference-generated golden vectors
    (visible_pos, corners_velo) in tests/test_oracle_golden.py."""
    corners_cam0 = np.asarray(corners_cam0, dtype=np.float64).reshape(-1, 8, 3)
    B = corners_cam0.shape[0]
    visible = np.zeros(B, bool)
    velo = np.zeros((B, 8, 3))
    TrCamToVelo = np.linalg.inv(TrVeloToCam)
    for b in range(B):
        corners = corners_cam0[b]
        u, v, depth = cam2image(K3, corners.T.copy())
        in_front = depth > 0.1
        in_image = (u >= 0) & (u < W) & (v >= 0) & (v < H)
        visible[b] = np.sum(in_front & in_image) >= 2
        corners_h = np.hstack([corners, np.ones((8, 1))])
        velo[b] = (TrCamToVelo @ corners_h.T).T[:, :3]
    return visible, velo


def cv2_resize_linear_u8(src, W, H):
    """``cv2.resize(src_u8, (W, H))`` with the default INTER_LINEAR on 8-bit data (V3:222: ``cv2.resize(mask.astype(np.uint8),
    (camera.width, camera.height))``), restated.  THIRD-PARTY ALGORITHM, PINNED BY CONSTRUCTION ONLY: OpenCV is not installed in this
    image and the reference holds no fixture of a resized mask; this follows OpenCV 4.x ``modules/imgproc/src/resize.cpp`` --
    ``resizeGeneric_`` with ``HResizeLinear<uchar, int, short, 2048>`` and ``VResizeLinear<uchar, int, short,
    FixedPtCast<int, uchar, 22>>`` (INTER_RESIZE_COEF_BITS = 11) -- as its C++ reference path computes it:
      * per destination column dx: ``fx = float((dx + 0.5) * scale_x - 0.5)`` with ``scale_x = 1 / (double(W) / w)``;
        ``sx = floor(fx)``, ``fx -= sx``; ``sx < 0 -> sx = 0, fx = 0``; ``sx >= w - 1 -> sx = w - 1, fx = 0``;
        weights ``a0 = saturate_short(round_half_even((1 - fx) * 2048))``, ``a1 = saturate_short(round_half_even(fx * 2048))``;
      * per destination row dy: fy, sy, b0, b1 in the same way BUT WITHOUT the two clamps -- ``resize()`` only clamps in its x loop;
        the row loop of ``resizeGeneric_Invoker`` keeps the fraction and clips the two ROW INDICES instead
        (``clip(sy + k, 0, h)``), so a destination row above the first / below the last source row blends that row with itself
        under both weights: ``((b0 * t) >> 16) + ((b1 * t) >> 16)``, whose two truncations are not those of ``(2048 * t) >> 16``;
      * horizontal pass, 32-bit: ``S = src[sx] * a0 + src[min(sx + 1, w - 1)] * a1`` (beyond the last source column: src[sx] * 2048);
      * vertical pass: ``dst = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2``.
    The exact 2 x 2 decimation (h == 2 H and w == 2 W) is not linear in OpenCV: ``resize()`` switches it to INTER_AREA (``if
    (interpolation == INTER_LINEAR && is_area_fast && iscale_x == 2 && iscale_y == 2) interpolation = INTER_AREA``), whose 8-bit
    one-channel 2 x 2 path (``ResizeAreaFastVec``: the scalar loop and the SIMD one compute the same) is
    ``(S[2x] + S[2x+1] + nextS[2x] + nextS[2x+1] + 2) >> 2``.  Not restated: whatever a SIMD / IPP build of OpenCV rounds
    differently in the linear case.  Equal sizes return the input (cv2.resize copies)."""
    src = np.ascontiguousarray(src, dtype=np.uint8)
    h, w = src.shape
    if (h, w) == (H, W):
        return src.copy()
    if w == 2 * W and h == 2 * H:                                         # INTER_AREA's 2 x 2 fast path
        s = src.astype(np.int64)
        return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)

    def coeffs(n_dst, n_src, clamp):
        scale = 1.0 / (float(n_dst) / float(n_src))                       # double, as resize() computes it
        d = np.arange(n_dst, dtype=np.float64)
        f = ((d + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(np.float32)).astype(np.float32)
        if clamp:                                                         # the x loop of resize(): index AND fraction
            lo = s < 0
            s[lo] = 0; f[lo] = 0.0
            hi = s >= n_src - 1
            s[hi] = n_src - 1; f[hi] = 0.0
        c0 = np.clip(np.rint((np.float32(1.0) - f) * np.float32(2048.0)), -32768, 32767).astype(np.int64)
        c1 = np.clip(np.rint(f * np.float32(2048.0)), -32768, 32767).astype(np.int64)
        # (rows: resizeGeneric_Invoker clips the indices only -- clip(sy + k, 0, h))
        return np.clip(s, 0, n_src - 1), np.clip(s + 1, 0, n_src - 1), c0, c1

    sx, sx1, a0, a1 = coeffs(W, w, True)
    sy, sy1, b0, b1 = coeffs(H, h, False)
    s = src.astype(np.int64)
    rows = s[:, sx] * a0[None, :] + s[:, sx1] * a1[None, :]                  # [h, W] 32-bit sums (<= 255 * 2048)
    S0, S1 = rows[sy], rows[sy1]                                             # [H, W]
    out = (((b0[:, None] * (S0 >> 4)) >> 16) + ((b1[:, None] * (S1 >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)


def cv2_erode_cross_u8(src, iterations=1):
    """``cv2.erode(src_u8, cv2.getStructuringElement(cv2.MORPH_ELLIPSE, (3, 3)), iterations=iterations)`` on 8-bit VALUES (V3:83-90),
    restated: OpenCV's 3 x 3 ellipse is the cross [[0,1,0],[1,1,1],[0,1,0]]; erosion is the minimum over it; the default border value
    of erode is +infinity, i.e. pixels outside the image do not take part.  THIRD-PARTY ALGORITHM, PINNED BY CONSTRUCTION ONLY (OpenCV
    is not installed here; hand-made known answers in tests/test_oracle_masks.py)."""
    a = np.ascontiguousarray(src, dtype=np.uint8)
    for _ in range(int(iterations)):
        p = np.pad(a, 1, constant_values=255)
        a = np.minimum.reduce([p[1:-1, 1:-1], p[:-2, 1:-1], p[2:, 1:-1], p[1:-1, :-2], p[1:-1, 2:]])
    return a


def v3_masks_at_camera_size(masks, W, H, erosion_iterations=1):
    """What V3 makes of detector masks [M,h,w] (float 0..1) that do NOT arrive at the camera's size, statement by statement:
    ``(mask * 255).astype(np.uint8)`` -> ``cv2.erode(...)`` at the mask's own size -> ``.astype(np.float32) / 255.0`` (V3:82-97), then in
    extract_car_points_by_mask ``cv2.resize(mask.astype(np.uint8), (W, H))`` and ``> 0.5`` (V3:222-225).  Returns uint8 [M,H,W], nonzero
    = member."""
    out = []
    for m in np.asarray(masks):
        u8 = (np.asarray(m, dtype=np.float32) * 255).astype(np.uint8)
        er = cv2_erode_cross_u8(u8, erosion_iterations)
        back = er.astype(np.float32) / 255.0
        out.append((cv2_resize_linear_u8(back.astype(np.uint8), W, H) > 0.5).astype(np.uint8))
    return np.stack(out) if out else np.zeros((0, H, W), np.uint8)

