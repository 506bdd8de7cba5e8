"""cv2.resize(mask.astype(np.uint8), (W, H)) for masks that do not arrive at the camera's size (V3:222).  OpenCV is not installed
here and the reference holds no resized fixture: the restatement (oracle/numpy_path.py: cv2_resize_linear_u8, from OpenCV 4.x
resize.cpp) is PINNED BY CONSTRUCTION ONLY -- against values derived by hand from its formula and against two small results that
OpenCV's documentation and countless bug reports quote -- and the HIP kernel is compared with it bit for bit."""
import numpy as np
import pytest

from oracle import numpy_path as npp


def test_restatement_on_hand_derived_values():
    # [0, 255] -> 4 wide: sample positions 0.25 / 0.75 between the two pixels: 63.75 -> 64, 191.25 -> 191; clamped ends
    assert npp.cv2_resize_linear_u8(np.array([[0, 255]], np.uint8), 4, 1).tolist() == [[0, 64, 191, 255]]
    # the 2 x 2 checker that OpenCV users post: separable, so every row / column is the line above or a blend of two
    want = [[0, 64, 191, 255], [64, 96, 159, 191], [191, 159, 96, 64], [255, 191, 64, 0]]
    assert npp.cv2_resize_linear_u8(np.array([[0, 255], [255, 0]], np.uint8), 4, 4).tolist() == want
    # a 0 / 1 mask doubled: weights 0.25 / 0.75 -> 0.75 rounds to 1, 0.25 to 0: the region grows to exactly twice its size
    m = np.zeros((4, 6), np.uint8); m[1:3, 2:5] = 1
    out = npp.cv2_resize_linear_u8(m, 12, 8)
    want = np.zeros((8, 12), np.uint8); want[2:6, 4:10] = 1
    assert np.array_equal(out, want)
    # by hand, one pixel of a 3 x 4 -> 2 x 3 reduction: dx = 1: fx = 1.5 * (4/3) - 0.5 = 1.5 -> sx = 1, weights 1024 / 1024;
    # dy = 0: fy = 0.5 * 1.5 - 0.5 = 0.25 -> sy = 0, weights 1536 / 512
    src = (np.arange(12, dtype=np.uint8).reshape(3, 4) * 20)
    S0 = 20 * 1024 + 40 * 1024; S1 = 100 * 1024 + 120 * 1024
    px = (((1536 * (S0 >> 4)) >> 16) + ((512 * (S1 >> 4)) >> 16) + 2) >> 2
    assert npp.cv2_resize_linear_u8(src, 3, 2)[0, 1] == px == 50
    # a TOP-ROW edge pixel, by hand (the row loop of resizeGeneric_Invoker keeps the fraction and clips the row INDICES: dy = 0 of a
    # doubling has fy = -0.25 -> sy = -1, fy = 0.75, rows clip(-1) = clip(0) = 0 under BOTH weights 512 and 1536).  x tripled: dx = 3 has
    # fx = 3.5 / 3 - 0.5 = 0.667 -> a1 = round(0.667 * 2048) = 1365, so S = 1 * 1365, S >> 4 = 85:
    # ((512 * 85) >> 16) + ((1536 * 85) >> 16) + 2 = 0 + 1 + 2 -> >> 2 = 0, where one weight of 2048 would give (2 + 2) >> 2 = 1
    edge = np.array([[0, 1], [0, 1]], np.uint8)
    assert ((((512 * (1365 >> 4)) >> 16) + ((1536 * (1365 >> 4)) >> 16) + 2) >> 2) == 0 and ((((2048 * (1365 >> 4)) >> 16) + 2) >> 2) == 1
    got = npp.cv2_resize_linear_u8(edge, 6, 4)
    assert got[:, 3].tolist() == [0, 0, 0, 0]           # (rows 1, 2 blend the two equal source rows under 1536 / 512: the same sum; a fraction
                                                        #  clamped to 0 in rows 0 and 3 -- the x loop's rule -- would give [1, 0, 0, 1])
    # equal sizes: a copy
    assert np.array_equal(npp.cv2_resize_linear_u8(src, 4, 3), src)
    # an exact 2 x 2 decimation is the case OpenCV hands to INTER_AREA: the rounded mean of each 2 x 2 block, (sum + 2) >> 2 --
    # by hand: [[1, 2], [3, 4]] -> (10 + 2) >> 2 = 3; [[0, 1], [0, 0]] -> 3 >> 2 = 0; [[1, 1], [0, 0]] -> 4 >> 2 = 1;
    # [[255, 255], [255, 254]] -> 1021 >> 2 = 255
    blk = np.array([[1, 2, 0, 1, 1, 1, 255, 255], [3, 4, 0, 0, 0, 0, 255, 254]], np.uint8)
    assert npp.cv2_resize_linear_u8(blk, 4, 1).tolist() == [[3, 0, 1, 255]]
    # a 0 / 1 mask halved: a block survives iff at least two of its four pixels are set
    m = np.array([[1, 0, 1, 1, 0, 0], [0, 0, 0, 0, 0, 1], [1, 1, 1, 0, 1, 1], [1, 1, 0, 1, 1, 0]], np.uint8)
    assert npp.cv2_resize_linear_u8(m, 3, 2).tolist() == [[0, 1, 0], [1, 1, 1]]
    # twice in ONE axis only stays linear (is_area_fast also needs iscale_y == 2)
    assert npp.cv2_resize_linear_u8(np.array([[0, 255, 255, 0]], np.uint8), 2, 1).tolist() == [[128, 128]]
    # (the hand-over does not change a single byte: at an exact halving the linear path samples every block's centre with weights
    #  1024 / 1024 in both axes, ((1024 * (((a + b) * 1024) >> 4)) >> 16) = a + b without loss, so it is (a + b + c + d + 2) >> 2 too --
    #  checked here on random bytes by forcing the linear formula through a 2 x 2 tiling of the same image at one-axis halvings)
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (12, 20)).astype(np.uint8)
    s = img.astype(np.int64)
    horiz = ((s[:, 0::2] + s[:, 1::2]) * 1024)                                       # the linear horizontal pass at fx = 0.5
    lin = (((1024 * (horiz[0::2] >> 4)) >> 16) + ((1024 * (horiz[1::2] >> 4)) >> 16) + 2) >> 2
    assert np.array_equal(npp.cv2_resize_linear_u8(img, 10, 6), lin.astype(np.uint8))


@pytest.mark.gpu
@pytest.mark.parametrize("hw", [(188, 704), (104, 384), (160, 608), (376, 1408), (500, 1900), (1, 1), (3, 2000), (533, 17), (200, 1408),
                                (752, 2816), (752, 1408), (376, 2816)])      # (twice the camera in both axes: INTER_AREA; in one axis: linear)
@pytest.mark.parametrize("where", ["host", "device"])
def test_kernel_equals_the_restatement(calib, hw, where):
    import torch
    from lidar_object_detection_amd import synthetic as S
    from lidar_object_detection_amd._native import LpfContext
    _, T, K, W, H = S.default_calibration(calib)
    h, w = hw
    rng = np.random.default_rng(h * 7919 + w)
    planes = np.stack([(rng.random((h, w)) < 0.4).astype(np.uint8),                       # a 0 / 1 mask
                       rng.integers(0, 256, (h, w)).astype(np.uint8),                     # any bytes
                       np.full((h, w), 255, np.uint8)])
    with LpfContext(0) as ctx:
        ctx.set_camera(T, K, W, H, 0.0, 50.0)
        if where == "host":
            out = ctx.resize_masks(planes)
        else:
            out = ctx.resize_masks(torch.from_numpy(planes).to(torch.device("cuda", 0))).cpu().numpy()
    assert out.shape == (3, H, W) and out.dtype == np.uint8
    for a, p in zip(out, planes):
        assert np.array_equal(a, npp.cv2_resize_linear_u8(p, W, H))


@pytest.mark.gpu
def test_masks_of_another_size_through_the_reference_shaped_calls(calib):
    """run_frames and extract_car_points_by_mask with masks at the detector's own size == the same calls with the masks resized
    beforehand (by the restatement); float masks are cast as the reference casts them."""
    from lidar_object_detection_amd import pipeline
    from lidar_object_detection_amd import synthetic as S
    _, T, K, W, H = S.default_calibration(calib)
    cam = type("Cam", (), {"K": np.asarray(K), "width": W, "height": H})()
    sc = S.scene(120_000, n_masks=5, n_boxes=12, seed=8300, calib=calib)
    small = np.stack([m[::2, ::2] for m in sc["masks"]]).astype(np.float32)               # [5, H/2, W/2] float 0. / 1., as a detector's
    small[0] *= 0.9                                                                       # astype(uint8) -> 0: this mask vanishes
    full = np.stack([npp.cv2_resize_linear_u8(m.astype(np.uint8), W, H) for m in small])
    assert not full[0].any() and full[1].any()
    boxes3d = [{"corners_velo": c.tolist()} for c in sc["corners_velo"]]
    a = pipeline.run_frames([pipeline.FrameInputs(1, sc["points"], small, boxes3d, pipeline.default_colors(5))], T, cam, 50.0, 10, True)[0]
    b = pipeline.run_frames([pipeline.FrameInputs(1, sc["points"], full, boxes3d, pipeline.default_colors(5))], T, cam, 50.0, 10, True)[0]
    assert np.array_equal(a["count_mb"], b["count_mb"]) and np.array_equal(a["bg_assigned"], b["bg_assigned"]) and sum(len(x) for x in a["car_point_sets"]) > 100
    for x, y in zip(a["car_point_sets"], b["car_point_sets"]):
        assert np.array_equal(x, y)
    sa = pipeline.extract_car_points_by_mask(a["points_valid"], a["u_valid"], a["v_valid"], small, cam)
    sb = pipeline.extract_car_points_by_mask(a["points_valid"], a["u_valid"], a["v_valid"], full, cam)
    assert len(sa) == 5 and sum(len(x) for x in sa) > 100
    for x, y in zip(sa, sb):
        assert np.array_equal(x, y)


def test_value_erosion_restatement_on_hand_made_planes():
    """cv2.erode with the 3 x 3 ellipse (= cross) on 8-bit values: the minimum over the plus, the border left out (pinned by construction)"""
    a = np.array([[9, 9, 9, 9], [9, 5, 9, 9], [9, 9, 9, 0]], np.uint8)
    assert npp.cv2_erode_cross_u8(a, 1).tolist() == [[9, 5, 9, 9], [5, 5, 5, 0], [9, 5, 0, 0]]
    assert npp.cv2_erode_cross_u8(a, 0).tolist() == a.tolist()
    one = np.zeros((5, 5), np.uint8); one[1:4, 1:4] = 255
    want = np.zeros((5, 5), np.uint8); want[2, 2] = 255
    assert np.array_equal(npp.cv2_erode_cross_u8(one, 1), want) and not npp.cv2_erode_cross_u8(one, 2).any()
    full = np.full((3, 4), 255, np.uint8)
    assert np.array_equal(npp.cv2_erode_cross_u8(full, 3), full)            # the image border does not erode


@pytest.mark.gpu
@pytest.mark.parametrize("where", ["host", "device"])
@pytest.mark.parametrize("iters", [0, 1, 2, 3])
def test_erosion_kernel_equals_the_restatement(calib, where, iters):
    import torch
    from lidar_object_detection_amd import synthetic as S
    from lidar_object_detection_amd._native import LpfContext
    _, T, K, W, H = S.default_calibration(calib)
    rng = np.random.default_rng(40 + iters)
    planes = np.stack([(rng.random((94, 353)) < 0.7).astype(np.uint8) * 255, rng.integers(0, 256, (94, 353)).astype(np.uint8), np.full((94, 353), 7, np.uint8)])
    with LpfContext(0) as ctx:
        ctx.set_camera(T, K, W, H, 0.0, 50.0)
        out = ctx.erode_masks(planes, iters) if where == "host" else ctx.erode_masks(torch.from_numpy(planes).to(torch.device("cuda", 0)), iters).cpu().numpy()
    for a, p in zip(out, planes):
        assert np.array_equal(a, npp.cv2_erode_cross_u8(p, iters))


@pytest.mark.gpu
@pytest.mark.parametrize("where", ["host", "device"])
def test_v3_erosion_block_on_masks_of_another_size(calib, where):
    """V3's own order for detector masks that do not arrive at camera size: (mask * 255).astype(uint8) -> cv2.erode at the MASK's size
    -> / 255.0 (V3:82-97), then astype(uint8) + cv2.resize + > 0.5 in extract_car_points_by_mask (V3:222-225).  run_frames with
    erode_iters=1, v3_pipeline=True on such masks == run_frames on the masks the restated chain produces (host and device masks; a batch
    that mixes them with masks at camera size erodes those by the same chain)."""
    import torch
    from lidar_object_detection_amd import pipeline
    from lidar_object_detection_amd import synthetic as S
    _, T, K, W, H = S.default_calibration(calib)
    cam = type("Cam", (), {"K": np.asarray(K), "width": W, "height": H})()
    sc = S.scene(120_000, n_masks=5, n_boxes=12, seed=8300, calib=calib)
    small = np.stack([m[::2, ::2] for m in sc["masks"]]).astype(np.float32)               # [5, H/2, W/2] float 0. / 1.
    small[1] *= 0.999                                                                     # (mask * 255) -> 254: never 255 after the erosion's cast back
    want_small = npp.v3_masks_at_camera_size(small, W, H, 1)
    want_full = npp.v3_masks_at_camera_size(sc["masks"].astype(np.float32), W, H, 1)      # masks at camera size through the same statements
    assert not want_small[1].any() and want_small[0].any() and (want_small[0] != npp.cv2_resize_linear_u8(small[0].astype(np.uint8), W, H)).any()
    boxes3d = [{"corners_velo": c.tolist()} for c in sc["corners_velo"]]
    to = (lambda a: torch.from_numpy(a).to(torch.device("cuda", 0))) if where == "device" else (lambda a: a)
    items = [pipeline.FrameInputs(1, sc["points"], to(small), boxes3d, pipeline.default_colors(5))]
    ref = [pipeline.FrameInputs(1, sc["points"], want_small, boxes3d, pipeline.default_colors(5))]
    if where == "host":                                                                   # (a mixed batch: ragged device batches are refused elsewhere)
        items.append(pipeline.FrameInputs(2, sc["points"], sc["masks"].astype(np.float32), boxes3d, pipeline.default_colors(5)))
        ref.append(pipeline.FrameInputs(2, sc["points"], want_full, boxes3d, pipeline.default_colors(5)))
    a = pipeline.run_frames(items, T, cam, 50.0, 10, True, erode_iters=1, v3_pipeline=True)
    b = pipeline.run_frames(ref, T, cam, 50.0, 10, True)
    for x, y in zip(a, b):
        assert np.array_equal(x["count_mb"], y["count_mb"]) and np.array_equal(x["bg_assigned"], y["bg_assigned"]) and sum(len(s_) for s_ in x["car_point_sets"]) > 100
        for s1, s2 in zip(x["car_point_sets"], y["car_point_sets"]):
            assert np.array_equal(s1, s2)


@pytest.mark.gpu
def test_same_color_indexes_a_mask_at_its_own_size(calib):
    """Same_color.py:124: ``y < mask.shape[0] and x < mask.shape[1] and mask[y, x] > 0.5`` -- no resize; the first mask that matches
    wins.  label_points_first_match with masks smaller / larger than the camera image == that loop, literally."""
    from lidar_object_detection_amd import pipeline
    from lidar_object_detection_amd import synthetic as S
    _, T, K, W, H = S.default_calibration(calib)
    cam = type("Cam", (), {"K": np.asarray(K), "width": W, "height": H})()
    sc = S.scene(60_000, n_masks=3, n_boxes=2, seed=8301, calib=calib)
    rng = np.random.default_rng(3)
    masks = [(rng.random((200, 900)) < 0.3).astype(np.float32), (rng.random((500, 1500)) < 0.3).astype(np.float32), sc["masks"][2].astype(np.float32)]
    got = pipeline.label_points_first_match(sc["points"], T, cam, masks, depth_max=30.0)
    u, v, _, valid_indices = pipeline.project_points(sc["points"], T, cam, depth_max=30.0, want_depth=False)
    car, first, bg = [], [], []
    for idx in valid_indices:
        x, y = u[idx], v[idx]
        for i, mk in enumerate(masks):
            if y < mk.shape[0] and x < mk.shape[1] and mk[y, x] > 0.5:
                car.append(idx); first.append(i)
                break
        else:
            bg.append(idx)
    assert len(car) > 500 and len(set(first)) == 3
    assert np.array_equal(got["car_idx"], car) and np.array_equal(got["car_mask"], first) and np.array_equal(got["background_idx"], bg)


@pytest.mark.gpu
def test_masks_produced_on_a_side_stream_just_before_the_call(calib):
    """Ordering (include/lpf.h, "Ordering contract"): masks that a long chain of kernels on ANOTHER torch stream is still
    producing when resize_masks / erode_masks are called, results read back on that stream with no host-side wait in between: the
    context draws an edge in from torch's current stream and an edge out to it."""
    import torch
    from lidar_object_detection_amd import synthetic as S
    from lidar_object_detection_amd._native import LpfContext
    _, T, K, W, H = S.default_calibration(calib)
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(77)
    planes = (rng.random((4, 188, 704)) < 0.4).astype(np.uint8)
    base = torch.from_numpy(planes).to(dev)
    big = torch.rand((2048, 2048), device=dev)
    torch.cuda.synchronize(dev)
    side = torch.cuda.Stream(dev)
    with LpfContext(0) as ctx:
        ctx.set_camera(T, K, W, H, 0.0, 50.0)
        ctx.resize_masks(base); ctx.erode_masks(base, 1)       # (allocations, outside the ordered region)
        torch.cuda.synchronize(dev)
        with torch.cuda.stream(side):
            x = big
            for _ in range(40):                             # a few milliseconds of work in front of the masks
                x = (x @ big) * 1e-3
            late = (base.to(torch.float32) + (x.sum() * 0.0)).to(torch.uint8)      # the masks: the last link of the chain
            er = ctx.erode_masks(late.contiguous(), 1)
            out = ctx.resize_masks(er)
            got_er, got = er.cpu().numpy(), out.cpu().numpy()                       # (copies queued on the side stream)
    for a, b, p in zip(got_er, got, planes):
        want_er = npp.cv2_erode_cross_u8(p, 1)
        assert np.array_equal(a, want_er)
        assert np.array_equal(b, npp.cv2_resize_linear_u8(want_er, W, H))
