"""Deterministic synthetic scenes in the shapes BASELINE.json's configs name.

There is no network and the KITTI-360 sample cannot travel to the GPU box, so the
benchmark and the randomised parity tests draw clouds, masks and boxes from seeded
generators with the real camera-0 calibration of the sample.
"""
import numpy as np

# KITTI-360 sample, camera 0 (KITTI360_sample/calibration/*.txt, composed as V3:527-535).
# Kept as literals so the generators work where the dataset is absent; the unit tests
# compare them with tests/golden/calib_cam0.npz.
K_CAM0 = np.array([[552.554261, 0.0, 682.049453],
                   [0.0, 552.554261, 238.769549],
                   [0.0, 0.0, 1.0]])
WIDTH, HEIGHT = 1408, 376
TR_VELO_TO_CAM = np.array([
    [0.04307104361409124, -0.9990043710100727, -0.01162548558261251, 0.26234696454785045],
    [-0.0882928649736657, 0.007784614038872, -0.9960641393796893, -0.10763413732850417],
    [0.9951629289402039, 0.04392796941355016, -0.08786966658919294, -0.8292052503750128],
    [0.0, 0.0, 0.0, 1.0]])
R_RECT0 = np.array([[0.999974, -0.007141, -0.000089, 0.0],
                    [0.007141, 0.999969, -0.003247, 0.0],
                    [0.000112, 0.003247, 0.999995, 0.0],
                    [0.0, 0.0, 0.0, 1.0]])

# (height, width, length) coefficients of the 8 corners in the sample's JSON order
_CORNER_HWL = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0],
                        [1, 0, 1], [0, 0, 1], [1, 1, 1], [0, 1, 1]], dtype=np.float64)


def default_calibration(calib=None):
    """(TrVeloToCam, TrVeloToRect, K3x3, W, H); ``calib`` may be the golden calib npz dict."""
    if calib is not None:
        return (np.asarray(calib["TrVeloToCam"]), np.asarray(calib["TrVeloToRect"]),
                np.asarray(calib["K"])[:, :3], int(calib["width"]), int(calib["height"]))
    return TR_VELO_TO_CAM, R_RECT0 @ TR_VELO_TO_CAM, K_CAM0, WIDTH, HEIGHT


def synthetic_cloud(n, seed=0):
    """f32 [n,4]: x, y ~ U(-80, 80), z ~ U(-3, 3), reflectance ~ U(0, 1)."""
    rng = np.random.default_rng(seed)
    pts = np.empty((n, 4), np.float32)
    pts[:, 0] = rng.uniform(-80.0, 80.0, n)
    pts[:, 1] = rng.uniform(-80.0, 80.0, n)
    pts[:, 2] = rng.uniform(-3.0, 3.0, n)
    pts[:, 3] = rng.uniform(0.0, 1.0, n)
    return pts


def synthetic_disk_masks(m, seed=0, width=WIDTH, height=HEIGHT):
    """m random disk masks (uint8 [m,H,W], radius U(20,120) px) + their xyxy rectangles."""
    rng = np.random.default_rng(seed + 1)
    yy, xx = np.mgrid[0:height, 0:width]
    masks = np.zeros((m, height, width), np.uint8)
    boxes = np.zeros((m, 4), np.float32)
    for i in range(m):
        r = rng.uniform(20.0, 120.0)
        cx, cy = rng.uniform(0, width), rng.uniform(0, height)
        masks[i] = ((xx - cx) ** 2 + (yy - cy) ** 2) <= r * r
        boxes[i] = (max(cx - r, 0), max(cy - r, 0), min(cx + r, width - 1), min(cy + r, height - 1))
    return masks, boxes


def synthetic_boxes(b, seed=0, velo_to_cam=None):
    """b random car-sized boxes: (corners_cam0 [b,8,3], corners_velo [b,8,3]) in the JSON corner order."""
    rng = np.random.default_rng(seed + 2)
    velo_to_cam = TR_VELO_TO_CAM if velo_to_cam is None else velo_to_cam
    cam_to_velo = np.linalg.inv(velo_to_cam)
    cam = np.zeros((b, 8, 3))
    for i in range(b):
        centre = np.array([rng.uniform(-20, 20), rng.uniform(0.5, 1.5), rng.uniform(5, 45)])
        h, w, l = np.array([1.65, 1.97, 4.43]) * rng.uniform(0.8, 1.2, 3)
        yaw = rng.uniform(0, 2 * np.pi)
        eh = np.array([0.0, 1.0, 0.0]) * h
        ew = np.array([np.cos(yaw), 0.0, np.sin(yaw)]) * w
        el = np.array([-np.sin(yaw), 0.0, np.cos(yaw)]) * l
        c0 = centre - 0.5 * (eh + ew + el)
        cam[i] = c0 + _CORNER_HWL @ np.stack([eh, ew, el])
    homo = np.concatenate([cam.reshape(-1, 3), np.ones((b * 8, 1))], axis=1)
    velo = (cam_to_velo @ homo.T).T[:, :3].reshape(b, 8, 3)
    return cam, velo


def scene(n_points, n_masks=8, n_boxes=32, seed=0, calib=None):
    """One synthetic frame: dict(points, masks, boxes2d, corners_cam0, corners_velo)."""
    velo_to_cam, _, _, w, h = default_calibration(calib)
    masks, boxes2d = synthetic_disk_masks(n_masks, seed, w, h)
    cam, velo = synthetic_boxes(n_boxes, seed, velo_to_cam)
    return dict(points=synthetic_cloud(n_points, seed), masks=masks, boxes2d=boxes2d,
                corners_cam0=cam, corners_velo=velo)
