"""ctypes front-end of oracle/lpf_oracle.c (TEST INFRASTRUCTURE ONLY).

Loaded by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the
product package must never import this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("LPF_ORACLE_SO") or os.path.join(_HERE, "liblpf_oracle.so")      # (the sanitizer build: tests/test_host_sanitized.py)
_lib = None

_P = ctypes.c_void_p
_I64 = ctypes.c_int64


def build(force=False):
    """Compile the C restatement with gcc (seconds)."""
    src = os.path.join(_HERE, "lpf_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.orc_run.restype = ctypes.c_int
        _lib.orc_pack_masks.restype = ctypes.c_int
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(_P)


def _pts4(points):
    p = np.ascontiguousarray(points, dtype=np.float32)
    assert p.ndim == 2 and p.shape[1] == 4, p.shape
    return p


def project(points, T, K):
    """K1+K2 for f32[N,4] points -> dict(u64, v64, u32, v32, depth, uf, vf)."""
    p = _pts4(points)
    n = p.shape[0]
    T = np.ascontiguousarray(T, dtype=np.float64).reshape(16)
    K = np.ascontiguousarray(K, dtype=np.float64).reshape(9)
    o = dict(u64=np.empty(n, np.int64), v64=np.empty(n, np.int64),
             u32=np.empty(n, np.int32), v32=np.empty(n, np.int32),
             depth=np.empty(n, np.float64), uf=np.empty(n, np.float64), vf=np.empty(n, np.float64))
    lib().orc_project(_ptr(p), _I64(n), _ptr(T), _ptr(K), _ptr(o["u64"]), _ptr(o["v64"]),
                      _ptr(o["u32"]), _ptr(o["v32"]), _ptr(o["depth"]), _ptr(o["uf"]), _ptr(o["vf"]))
    return o


def binarize_f32(masks, v3_erosion):
    m = np.ascontiguousarray(masks, dtype=np.float32)
    out = np.empty(m.shape, np.uint8)
    lib().orc_binarize_f32(_ptr(m), _I64(m.size), ctypes.c_int(int(v3_erosion)), _ptr(out))
    return out


def erode_cross3(img):
    a = np.ascontiguousarray(img, dtype=np.uint8)
    assert a.ndim == 2
    out = np.empty_like(a)
    lib().orc_erode_cross3(_ptr(a), _ptr(out), ctypes.c_int(a.shape[0]), ctypes.c_int(a.shape[1]))
    return out


def pack_masks(masks_u8, erode_iters=0, H=None, W=None):
    """u8[M,H,W] (nonzero = member) -> u32[H,W] label image (bit m = mask m)."""
    m = np.ascontiguousarray(masks_u8, dtype=np.uint8)
    if m.ndim == 2:
        m = m[None]
    if m.shape[0] == 0:
        return np.zeros((H, W), np.uint32)
    M, H, W = m.shape
    lab = np.empty((H, W), np.uint32)
    rc = lib().orc_pack_masks(_ptr(m), ctypes.c_int(M), ctypes.c_int(H), ctypes.c_int(W),
                              ctypes.c_int(erode_iters), _ptr(lab))
    if rc != 0:
        raise ValueError("orc_pack_masks rc=%d" % rc)
    return lab


def points_in_box(points_xyz, corners, oriented=True):
    """inside mask of f32[k,3] (or [k,4]) points against one f64[8,3] box."""
    p = np.ascontiguousarray(points_xyz, dtype=np.float32)
    if p.size == 0:
        return np.zeros(0, bool)
    c = np.ascontiguousarray(corners, dtype=np.float64).reshape(24)
    out = np.empty(p.shape[0], np.uint8)
    lib().orc_points_in_box(_ptr(p), _I64(p.shape[0]), ctypes.c_int(p.shape[1]), _ptr(c),
                            ctypes.c_int(int(oriented)), _ptr(out))
    return out.astype(bool)


def depth_image(points, T, K, W, H, dmin, dmax):
    """(D f64[H,W], winner int32[H,W]): last valid point per pixel (seg_with_pointcloud.py:160-170)."""
    p = _pts4(points)
    T = np.ascontiguousarray(T, dtype=np.float64).reshape(16)
    K = np.ascontiguousarray(K, dtype=np.float64).reshape(9)
    D, win = np.empty((H, W), np.float64), np.empty((H, W), np.int32)
    lib().orc_depth_image(_ptr(p), _I64(p.shape[0]), _ptr(T), _ptr(K), ctypes.c_int(W), ctypes.c_int(H),
                          ctypes.c_double(dmin), ctypes.c_double(dmax), _ptr(D), _ptr(win))
    return D, win


def run(points, T, K, W, H, dmin, dmax, label_img=None, M=0, corners=None, oriented=True,
        want_float=True, inst_stride=None):
    """The whole per-frame path on the CPU; mirrors LpfContext.run()'s outputs."""
    p = _pts4(points)
    n = p.shape[0]
    T = np.ascontiguousarray(T, dtype=np.float64).reshape(16)
    K = np.ascontiguousarray(K, dtype=np.float64).reshape(9)
    if corners is None:
        corners = np.zeros((0, 8, 3))
    c = np.ascontiguousarray(corners, dtype=np.float64).reshape(-1, 8, 3)
    B = c.shape[0]
    if label_img is not None:
        label_img = np.ascontiguousarray(label_img, dtype=np.uint32)
        assert label_img.shape == (H, W)
    else:
        M = 0
    if inst_stride is None:
        inst_stride = max(n, 1)
    o = dict(u=np.empty(n, np.int32), v=np.empty(n, np.int32), label_bits=np.empty(n, np.uint32),
             valid_idx=np.empty(n, np.int64), inst_idx=np.empty((M, inst_stride), np.int64),
             inst_count=np.zeros(M, np.int64), count_mb=np.zeros((M, B), np.int64),
             best_box=np.empty(M, np.int32), best_cnt=np.empty(M, np.int64))
    if want_float:
        o.update(depth=np.empty(n, np.float64), uf=np.empty(n, np.float64), vf=np.empty(n, np.float64))
    nv = _I64(0)
    rc = lib().orc_run(_ptr(p), _I64(n), _ptr(T), _ptr(K), ctypes.c_int(W), ctypes.c_int(H),
                       ctypes.c_double(dmin), ctypes.c_double(dmax),
                       _ptr(label_img), ctypes.c_int(M), _ptr(c), ctypes.c_int(B),
                       ctypes.c_int(int(oriented)),
                       _ptr(o["u"]), _ptr(o["v"]), _ptr(o.get("depth")), _ptr(o.get("uf")), _ptr(o.get("vf")),
                       _ptr(o["label_bits"]), _ptr(o["valid_idx"]), ctypes.byref(nv),
                       _ptr(o["inst_idx"]), _I64(inst_stride), _ptr(o["inst_count"]),
                       _ptr(o["count_mb"]), _ptr(o["best_box"]), _ptr(o["best_cnt"]))
    if rc != 0:
        raise RuntimeError("orc_run rc=%d" % rc)
    o["n_valid"] = int(nv.value)
    o["valid_idx"] = o["valid_idx"][:o["n_valid"]]
    o["inst_lists"] = [o["inst_idx"][m, :o["inst_count"][m]].copy() for m in range(M)]
    return o
