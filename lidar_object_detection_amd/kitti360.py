"""KITTI-360 dataset + calibration readers used either side of the hot path.

The reference imports these from the un-vendored ``kitti360scripts`` package
(``Coding_testes/V3_point_cloud_with_erosion.py:9-10``, used at ``V3:524-530`` and
``V3:568``); that package is not under /root/reference, so this module restates the
published file formats and the ``cam2image`` formula.  It is host
glue: the per-point arithmetic of ``cam2image`` for LiDAR points runs in the HIP
kernel, this NumPy form only serves the 8-corner box projections.
"""
import glob
import json
import os

import numpy as np


def _check_file(path):
    if not os.path.isfile(path):
        raise RuntimeError("%s does not exist!" % path)


def loadCalibrationRigid(filename):
    """3x4 row-major text -> 4x4 float64 (last row 0 0 0 1).  Ref use: V3:529."""
    _check_file(filename)
    vals = np.loadtxt(filename, dtype=np.float64).reshape(3, 4)
    return np.concatenate((vals, np.array([[0.0, 0.0, 0.0, 1.0]])))


def loadCalibrationCameraToPose(filename):
    """``image_0k: <12 floats>`` lines -> {name: 4x4 float64}.  Ref use: V3:530."""
    _check_file(filename)
    out = {}
    with open(filename) as f:
        for line in f:
            parts = line.split()
            if not parts or not parts[0].endswith(":"):
                continue
            name = parts[0][:-1]
            if name.startswith("image_") and len(parts) >= 13:
                m = np.array([float(x) for x in parts[1:13]], dtype=np.float64).reshape(3, 4)
                out[name] = np.concatenate((m, np.array([[0.0, 0.0, 0.0, 1.0]])))
    for cam in ("image_00", "image_01", "image_02", "image_03"):
        if cam not in out:
            raise RuntimeError("%s: no %s entry" % (filename, cam))
    return out


class CameraPerspective:
    """Rectified pinhole camera 0/1 of KITTI-360 (ref construction: V3:524).

    Attributes mirror what the reference touches: ``K`` (3x4 ``P_rect_0k``),
    ``R_rect`` (4x4), ``width``, ``height`` and ``cam2image``.
    """

    def __init__(self, root_dir, seq="2013_05_28_drive_0000_sync", cam_id=0, load_poses=False):
        if cam_id not in (0, 1):
            raise RuntimeError("perspective camera ids are 0 and 1")
        self.cam_id = cam_id
        calib_dir = os.path.join(root_dir, "calibration")
        self.intrinsic_file = os.path.join(calib_dir, "perspective.txt")
        self.pose_file = os.path.join(root_dir, "data_poses", seq, "poses.txt")
        self.camToPose = loadCalibrationCameraToPose(
            os.path.join(calib_dir, "calib_cam_to_pose.txt"))["image_%02d" % cam_id]
        self.load_intrinsics(self.intrinsic_file)
        self.cam2world = {}
        if load_poses and os.path.isfile(self.pose_file):
            poses = np.loadtxt(self.pose_file)
            for row in poses:
                pose = np.concatenate((row[1:].reshape(3, 4), np.array([[0.0, 0.0, 0.0, 1.0]])))
                self.cam2world[int(row[0])] = pose @ self.camToPose @ np.linalg.inv(self.R_rect)

    @classmethod
    def from_arrays(cls, K, R_rect, width, height, cam_id=0):
        """Synthetic camera (no files): used by benchmarks and tests."""
        self = cls.__new__(cls)
        self.cam_id = cam_id
        K = np.asarray(K, dtype=np.float64)
        self.K = np.concatenate((K[:3, :3], np.zeros((3, 1))), axis=1) if K.shape == (3, 3) else K.copy()
        self.R_rect = np.asarray(R_rect, dtype=np.float64).copy()
        self.width, self.height = int(width), int(height)
        self.camToPose = np.eye(4)
        self.cam2world = {}
        return self

    def load_intrinsics(self, intrinsic_file):
        _check_file(intrinsic_file)
        K = R_rect = None
        width = height = -1
        tag = "%02d:" % self.cam_id
        with open(intrinsic_file) as f:
            for line in f:
                parts = line.split()
                if not parts:
                    continue
                if parts[0] == "P_rect_" + tag:
                    K = np.array([float(x) for x in parts[1:13]], dtype=np.float64).reshape(3, 4)
                elif parts[0] == "R_rect_" + tag:
                    R_rect = np.eye(4)
                    R_rect[:3, :3] = np.array([float(x) for x in parts[1:10]], dtype=np.float64).reshape(3, 3)
                elif parts[0] == "S_rect_" + tag:
                    width, height = int(float(parts[1])), int(float(parts[2]))
        if K is None or R_rect is None or width <= 0 or height <= 0:
            raise RuntimeError("%s: incomplete intrinsics for camera %d" % (intrinsic_file, self.cam_id))
        self.K, self.R_rect, self.width, self.height = K, R_rect, width, height

    def cam2image(self, points):
        """points f64[3,N] (or [B,3,N]) in the rectified camera frame -> (u, v, depth).

        u, v are int64 (np.round half-to-even), depth is the signed third row with
        exact zeros replaced by -1e-6 -- the semantics the reference relies on at
        V3:130-134 and V3:568-584.
        """
        points = np.asarray(points, dtype=np.float64)
        ndim = points.ndim
        if ndim == 2:
            points = points[None]
        proj = np.matmul(self.K[:3, :3].reshape(1, 3, 3), points)
        depth = proj[:, 2, :]
        depth[depth == 0] = -1e-6
        with np.errstate(invalid="ignore", over="ignore"):
            u = np.round(proj[:, 0, :] / np.abs(depth)).astype(np.int64)
            v = np.round(proj[:, 1, :] / np.abs(depth)).astype(np.int64)
        if ndim == 2:
            return u[0], v[0], depth[0]
        return u, v, depth


class Kitti360Viewer3DRaw:
    """Velodyne scan reader (ref: V3:18-28): ``%010d.bin`` little-endian f32 [N,4]."""

    def __init__(self, seq=0, root_dir=None):
        root = root_dir if root_dir is not None else os.environ["KITTI360_DATASET"]
        sequence = "2013_05_28_drive_%04d_sync" % seq
        self.raw3DPcdPath = os.path.join(root, "data_3d_raw", sequence, "velodyne_points", "data")

    def available_frames(self):
        files = sorted(glob.glob(os.path.join(self.raw3DPcdPath, "*.bin")))
        return [int(os.path.basename(f).split(".")[0]) for f in files]

    def loadVelodyneData(self, frame=0):
        pcd_file = os.path.join(self.raw3DPcdPath, "%010d.bin" % frame)
        if not os.path.isfile(pcd_file):
            raise RuntimeError("%s does not exist!" % pcd_file)
        return np.fromfile(pcd_file, dtype=np.float32).reshape(-1, 4)


def load_bounding_boxes(json_path):
    """Box list of one frame, [] (plus the reference's message) when absent (V3:31-38)."""
    try:
        with open(json_path, "r") as f:
            return json.load(f)
    except FileNotFoundError:
        print("No bounding boxes found: %s" % json_path)
        return []


def velo_to_rect_transforms(root_dir, camera, cam_id=0):
    """(TrVeloToCam, TrVeloToRect) exactly as composed at V3:527-535."""
    calib = os.path.join(root_dir, "calibration")
    cam0_to_velo = loadCalibrationRigid(os.path.join(calib, "calib_cam_to_velo.txt"))
    cam_to_pose = loadCalibrationCameraToPose(os.path.join(calib, "calib_cam_to_pose.txt"))
    camk_to_cam0 = np.linalg.inv(cam_to_pose["image_00"]) @ cam_to_pose["image_%02d" % cam_id]
    cam_to_velo = cam0_to_velo @ camk_to_cam0
    velo_to_cam = np.linalg.inv(cam_to_velo)
    velo_to_rect = np.matmul(camera.R_rect, velo_to_cam)
    return velo_to_cam, velo_to_rect
