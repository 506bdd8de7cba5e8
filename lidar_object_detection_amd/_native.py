"""ctypes binding of liblpf.so (include/lpf.h) -- the only route to the hot path.

There is no CPU fallback: if the HIP library is missing or no GPU is present the
calls raise.  NumPy arrays go through the ABI's host-pointer mode; torch CUDA (ROCm)
tensors are passed by ``data_ptr()`` in device mode.
"""
import collections
import ctypes
import os
import sys
import weakref

import numpy as np

from . import _build

LPF_MAX_MASKS = 32
_P = ctypes.c_void_p
_I64 = ctypes.c_int64


class LpfError(RuntimeError):
    """An lpf_* call returned a negative status (message from lpf_last_error)."""

    def __init__(self, code, msg):
        super().__init__("liblpf error %d: %s" % (code, msg))
        self.code = code


class FrameSummary(ctypes.Structure):
    _fields_ = [("n_valid", _I64), ("n_labelled", _I64),
                ("inst_count", _I64 * LPF_MAX_MASKS), ("inst_off", _I64 * (LPF_MAX_MASKS + 1)),
                ("best_cnt", _I64 * LPF_MAX_MASKS), ("best_box", ctypes.c_int32 * LPF_MAX_MASKS),
                ("inst_overflow", ctypes.c_int32), ("reserved", ctypes.c_int32)]


SUMMARY_DTYPE = np.dtype([("n_valid", "<i8"), ("n_labelled", "<i8"), ("inst_count", "<i8", (32,)),
                          ("inst_off", "<i8", (33,)), ("best_cnt", "<i8", (32,)), ("best_box", "<i4", (32,)),
                          ("inst_overflow", "<i4"), ("reserved", "<i4")])
assert SUMMARY_DTYPE.itemsize == ctypes.sizeof(FrameSummary) == 928


class Outputs(ctypes.Structure):
    _fields_ = [("uv", _P), ("label_bits", _P), ("depth", _P), ("u_f", _P), ("v_f", _P),
                ("valid_idx", _P), ("inst_idx", _P), ("inst_cap", _I64), ("count_mb", _P),
                ("summary", _P), ("on_device", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("uv_valid", _P), ("label_valid", _P)]


class FrameJob(ctypes.Structure):
    """lpf_frame_job (include/lpf.h): one frame of a stream -- scan, masks + rectangles, cam-0 box corners, outputs -- for lpf_run_frame"""
    _fields_ = [("pts", _P), ("n_points", _I64), ("masks", _P), ("mask_rects", _P), ("corners_cam0", _P), ("T_cam_to_velo", _P),
                ("n_masks", ctypes.c_int32), ("n_boxes", ctypes.c_int32), ("filter_visible", ctypes.c_int32), ("oriented", ctypes.c_int32),
                ("out", Outputs)]


_libs = {}


def library_path():
    """liblpf.so of this package.  LPF_LIBRARY names another build of the same ABI, and is honoured ONLY in lab runs (LPF_LAB=1 in the
    environment, which `bench.py --lab ...` and the tools set): a stray variable must not silently swap the library a number or a
    test is made with."""
    env = os.environ.get("LPF_LIBRARY")
    if env:
        if os.environ.get("LPF_LAB") != "1":
            raise LpfError(-3, "LPF_LIBRARY=%s is set but this is not a lab run (LPF_LAB=1; bench.py --lab ...): refusing to load another "
                               "library in place of %s" % (env, _build.LIB))
        return env
    return _build.LIB


def _own(path):
    """lab flag if `path` is one of this package's two libraries (they are tied to the sources beside them), else None"""
    for lib, lab in ((_build.LIB, False), (_build.LAB_LIB, True)):
        if os.path.abspath(path) == os.path.abspath(lib):
            return lab
    return None


def load(path=None):
    """dlopen liblpf.so (or another build of the same ABI at ``path``).  The package's own libraries are tied to their sources: a
    missing or STALE one (its compiled-in build id differs from _build.source_id()) is rebuilt with hipcc, and refused if that is
    not possible -- never loaded as it is, and there is no CPU path to fall back to."""
    path = os.path.abspath(path or library_path())
    if path in _libs:
        return _libs[path]
    lab = _own(path)
    if lab is not None and _build.needs_build(path):
        have = _build.library_id(path)
        try:
            import fcntl
            with open(path + ".lock", "w") as lk:           # (the ranks of a multi-process run must not rebuild the file under each other)
                fcntl.flock(lk, fcntl.LOCK_EX)
                if _build.needs_build(path):
                    _build.build(lab=lab)
        except Exception as e:
            raise LpfError(-2, "%s is %s and cannot be rebuilt here (%s): run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(hipcc --offload-arch=gfx950); there is no CPU path" % (
                                   path, "missing" if not os.path.exists(path) else "stale (built from sources %s, these are %s)" % (have, _build.source_id(lab)), e))
    if not os.path.exists(path):
        raise LpfError(-2, "%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950); there is no CPU path" % path)
    lib = ctypes.CDLL(path)
    build_id = None
    if hasattr(lib, "lpf_build_id"):
        lib.lpf_build_id.restype = ctypes.c_char_p
        build_id = (lib.lpf_build_id() or b"").decode()
    if lab is not None and build_id != _build.source_id(lab):
        raise LpfError(-2, "%s reports build id %s, the sources beside it are %s: refusing a library that is not built from them" % (
            path, build_id, _build.source_id(lab)))
    lib._lpf_info = {"path": path, "build_id": build_id}
    if hasattr(lib, "lpf_host_alloc"):
        lib.lpf_host_alloc.restype = _P
        lib.lpf_host_alloc.argtypes = [ctypes.c_size_t]
        lib.lpf_host_free.restype = None
        lib.lpf_host_free.argtypes = [_P]
    lib.lpf_last_error.restype = ctypes.c_char_p
    lib.lpf_last_error.argtypes = [_P]
    lib.lpf_create.argtypes = [ctypes.POINTER(_P), ctypes.c_int]
    lib.lpf_destroy.argtypes = [_P]
    lib.lpf_destroy.restype = None
    lib.lpf_set_stream.argtypes = [_P, _P]
    lib.lpf_use_own_stream.argtypes = [_P]
    lib.lpf_wait_for_stream.argtypes = [_P, _P]
    lib.lpf_release_to_stream.argtypes = [_P, _P]
    lib.lpf_sync.argtypes = [_P]
    lib.lpf_set_pipelined.argtypes = [_P, ctypes.c_int]
    if hasattr(lib, "lpf_set_geometry"):                     # lab builds only (-DLPF_LAB)
        lib.lpf_set_geometry.argtypes = [_P, ctypes.c_int]
        lib.lpf_lab_role_clock.argtypes = [_P, _P, ctypes.c_int]
    lib.lpf_allreduce_metrics.argtypes = [_P, _P, ctypes.c_int, ctypes.c_int, _P]
    lib.lpf_set_camera.argtypes = [_P, _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double]
    lib.lpf_set_masks_u8.argtypes = [_P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.lpf_set_mask_rects.argtypes = [_P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.lpf_resize_masks_u8.argtypes = [_P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P, ctypes.c_int]
    lib.lpf_erode_masks_u8.argtypes = [_P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P, ctypes.c_int]
    lib.lpf_set_masks_f32.argtypes = [_P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.lpf_set_label_image.argtypes = [_P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.lpf_get_label_image.argtypes = [_P, _P, ctypes.c_int]
    lib.lpf_set_boxes.argtypes = [_P, _P, _P, ctypes.c_int, ctypes.c_int]
    lib.lpf_set_boxes_ex.argtypes = [_P, _P, ctypes.c_int, _P, ctypes.c_int, ctypes.c_int]
    lib.lpf_set_boxes_cam0.argtypes = [_P, _P, ctypes.c_int, _P, ctypes.c_int, _P, ctypes.c_int, ctypes.c_int, _P, _P, _P, _P]
    lib.lpf_run.argtypes = [_P, _P, _I64, ctypes.c_int, ctypes.POINTER(Outputs)]
    lib.lpf_run_batch.argtypes = [_P, _P, _P, ctypes.c_int, ctypes.c_int, ctypes.POINTER(Outputs)]
    lib.lpf_run_frame.argtypes = [_P, ctypes.POINTER(FrameJob)]
    lib.lpf_points_in_boxes.argtypes = [_P, _P, _I64, ctypes.c_int, _P, ctypes.c_int, ctypes.c_int, _P, ctypes.c_int]
    lib.lpf_depth_image.argtypes = [_P, _P, _I64, ctypes.c_int, _P, _P]
    lib.lpf_prepare_boxes.argtypes = [_P, _P, ctypes.c_int, _P, _P, _P, _P, _P]
    lib.lpf_graph_begin.argtypes = [_P]
    lib.lpf_graph_end.argtypes = [_P, ctypes.POINTER(_P)]
    lib.lpf_graph_launch.argtypes = [_P, _P]
    lib.lpf_graph_destroy.argtypes = [_P]
    lib.lpf_graph_destroy.restype = None
    lib.lpf_get_stats.argtypes = [_P, _P, ctypes.c_int, ctypes.c_int]
    lib.lpf_profile_enable.argtypes = [_P, ctypes.c_int]
    lib.lpf_profile_read.argtypes = [_P, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_I64), ctypes.c_int]
    lib.lpf_profile_overhead.argtypes = [_P, ctypes.POINTER(ctypes.c_double)]
    lib.lpf_reader_create.argtypes = [_P, ctypes.POINTER(_P), ctypes.c_int, _I64]
    lib.lpf_reader_submit.argtypes = [_P, ctypes.c_char_p]
    lib.lpf_reader_next.argtypes = [_P, ctypes.POINTER(_P), ctypes.POINTER(_P), ctypes.POINTER(_I64)]
    lib.lpf_reader_submit_frame.argtypes = [_P, ctypes.c_char_p, ctypes.c_char_p]
    lib.lpf_reader_boxes.argtypes = [_P, ctypes.POINTER(_P), ctypes.POINTER(_P), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    lib.lpf_parse_boxes_json.argtypes = [ctypes.c_char_p, _P, _P, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    lib.lpf_reader_wait.argtypes = [_P]
    lib.lpf_reader_destroy.argtypes = [_P]
    lib.lpf_reader_destroy.restype = None
    _libs[path] = lib
    return lib


EXPORTED = ("lpf_abi_version", "lpf_build_id", "lpf_host_alloc", "lpf_host_free", "lpf_create", "lpf_destroy", "lpf_last_error", "lpf_set_stream", "lpf_use_own_stream", "lpf_wait_for_stream",
            "lpf_release_to_stream", "lpf_sync",
            "lpf_set_pipelined", "lpf_allreduce_metrics",
            "lpf_set_camera", "lpf_set_masks_u8", "lpf_set_masks_f32", "lpf_set_mask_rects", "lpf_set_label_image",
            "lpf_get_label_image", "lpf_set_boxes", "lpf_set_boxes_ex", "lpf_set_boxes_cam0", "lpf_run", "lpf_run_batch", "lpf_run_frame",
            "lpf_points_in_boxes", "lpf_prepare_boxes", "lpf_depth_image", "lpf_resize_masks_u8", "lpf_erode_masks_u8", "lpf_get_stats", "lpf_profile_enable", "lpf_profile_read", "lpf_profile_overhead",
            "lpf_graph_begin", "lpf_graph_end", "lpf_graph_launch", "lpf_graph_destroy",
            "lpf_reader_create", "lpf_reader_submit", "lpf_reader_next", "lpf_reader_wait", "lpf_reader_destroy",
            "lpf_reader_submit_frame", "lpf_reader_boxes", "lpf_parse_boxes_json")

BOXES_PARSED, BOXES_ABSENT, BOXES_OTHER, BOXES_NONE = 0, 1, 2, 3          # enum lpf_boxes_state


def parse_boxes_file(path):
    """``(state, index int32[B], corners_cam0 f64[B,8,3])`` of a ``BBoxes_<frame>.json`` file, parsed by the library
    (lpf_parse_boxes_json: no GPU involved; the doubles are json.load's).  state: BOXES_PARSED, BOXES_ABSENT (no such file) or
    BOXES_OTHER -- the file is not the plain ``[{"index": int, "corners_cam0": 8 x 3 numbers}, ...]`` and was not interpreted (use
    json.load)."""
    lib = load()
    path = os.fspath(path)
    try:
        cap = os.path.getsize(path) // 64 + 1          # (a box is at least 92 characters of text)
    except OSError:
        cap = 0
    corners, index = np.empty((cap, 8, 3), np.float64), np.empty(cap, np.int32)
    n, st = ctypes.c_int(0), ctypes.c_int(0)
    rc = lib.lpf_parse_boxes_json(path.encode(), corners.ctypes.data if cap else None, index.ctypes.data if cap else None, cap,
                                  ctypes.byref(n), ctypes.byref(st))
    if rc != 0:
        raise LpfError(rc, "lpf_parse_boxes_json(%s): %d boxes, room for %d" % (path, n.value, cap))
    if st.value != BOXES_PARSED:
        return st.value, np.empty(0, np.int32), np.empty((0, 8, 3), np.float64)
    return BOXES_PARSED, index[:n.value], corners[:n.value]


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _dev_ptr(t, dtype_name=None):
    """data_ptr of a contiguous torch ROCm tensor (or None)."""
    if t is None:
        return None
    if not t.is_cuda or not t.is_contiguous():
        raise ValueError("device buffers must be contiguous torch tensors on the GPU")
    if dtype_name is not None and str(t.dtype) != "torch." + dtype_name:
        raise ValueError("expected torch.%s, got %s" % (dtype_name, t.dtype))
    return t.data_ptr()


class Scan:
    """One velodyne scan handed out by ScanReader: ``points`` is a float32 [N,4] NumPy view of the
    pinned host copy (what loadVelodyneData returns, V3:24-28), ``dev_ptr`` its copy in HBM.  Both
    are only valid until the reader's next scan is fetched; copy ``points`` to keep it."""
    __slots__ = ("path", "n", "points", "dev_ptr", "_reader", "_ticket", "boxes_state", "box_index", "boxes_cam0")

    def __init__(self, path, n, points, dev_ptr, reader, ticket, boxes_state=BOXES_NONE, box_index=None, boxes_cam0=None):
        self.path, self.n, self.points, self.dev_ptr, self._reader, self._ticket = path, n, points, dev_ptr, reader, ticket
        # the frame's box file when the reader was given one (ScanReader(box_paths=...)): BOXES_* and, when PARSED, COPIES of the
        # reader's arrays: int32 [B], float64 [B,8,3]
        self.boxes_state, self.box_index, self.boxes_cam0 = boxes_state, box_index, boxes_cam0

    def _check_live(self):
        if self._reader._ticket != self._ticket or self._reader._h is None:
            raise LpfError(-3, "this Scan's buffers were recycled (a later scan has been fetched from its reader)")


class ScanReader:
    """Iterator over velodyne .bin files with read-ahead: a native worker thread reads the next files
    into pinned memory and copies them to HBM while the current scan is processed (lpf_reader_*).
    A missing file raises LpfError('<path> does not exist!') for that scan, like V3:26-27."""

    def __init__(self, ctx, paths, n_buffers=3, max_points=1 << 21, box_paths=None):
        self._ctx, self._lib = ctx, ctx._lib
        self._h = None
        self._ticket = 0
        self._paths = [os.fspath(p) for p in paths]
        # box_paths[i]: the BBoxes_<frame>.json of scan i (or None) -- parsed by the reader's worker beside the scan
        self._box_paths = None if box_paths is None else [None if b is None else os.fspath(b) for b in box_paths]
        if self._box_paths is not None and len(self._box_paths) != len(self._paths):
            raise ValueError("box_paths: one entry per scan")
        self._next_submit = 0
        self._delivered = 0
        self._depth = int(n_buffers)
        h = _P()
        ctx._check(self._lib.lpf_reader_create(ctx._h, ctypes.byref(h), int(n_buffers), int(max_points)))
        self._h = h
        ctx._readers.add(self)                  # the context closes its readers before it goes away
        self._top_up()

    def _top_up(self):                         # keep n_buffers - 1 scans ahead of the consumer
        while self._next_submit < len(self._paths) and self._next_submit - self._delivered < self._depth:
            bp = self._box_paths[self._next_submit] if self._box_paths is not None else None
            self._ctx._check(self._lib.lpf_reader_submit_frame(self._h, self._paths[self._next_submit].encode(), bp.encode() if bp else None))
            self._next_submit += 1

    def __iter__(self):
        return self

    def __len__(self):
        return len(self._paths)

    def __next__(self):
        if self._h is None or self._delivered >= len(self._paths):
            raise StopIteration
        path = self._paths[self._delivered]
        d, hp, n = _P(), _P(), _I64(0)
        self._ticket += 1
        rc = self._lib.lpf_reader_next(self._h, ctypes.byref(d), ctypes.byref(hp), ctypes.byref(n))
        self._delivered += 1
        self._top_up()
        self._ctx._check(rc)
        cnt = int(n.value)
        if cnt:
            buf = (ctypes.c_float * (cnt * 4)).from_address(hp.value)
            pts = np.frombuffer(buf, dtype=np.float32).reshape(cnt, 4)
        else:
            pts = np.zeros((0, 4), np.float32)
        state, bidx, bcam = BOXES_NONE, None, None
        if self._box_paths is not None:
            pc, pi, nb, st = _P(), _P(), ctypes.c_int(0), ctypes.c_int(BOXES_NONE)
            self._ctx._check(self._lib.lpf_reader_boxes(self._h, ctypes.byref(pc), ctypes.byref(pi), ctypes.byref(nb), ctypes.byref(st)))
            state = int(st.value)
            if state == BOXES_PARSED:
                b = int(nb.value)
                if b:
                    bcam = np.frombuffer((ctypes.c_double * (b * 24)).from_address(pc.value), dtype=np.float64).reshape(b, 8, 3).copy()
                    bidx = np.frombuffer((ctypes.c_int32 * b).from_address(pi.value), dtype=np.int32).copy()
                else:
                    bcam, bidx = np.empty((0, 8, 3), np.float64), np.empty(0, np.int32)
        return Scan(path, cnt, pts, d.value, self, self._ticket, state, bidx, bcam)

    def wait(self):
        self._ctx._check(self._lib.lpf_reader_wait(self._h))

    def close(self):
        if self._h is not None:
            self._lib.lpf_reader_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LpfContext:
    """One context = one GPU + one stream (not thread-safe): the C ABI, object-shaped."""

    def __init__(self, device=0, library=None):
        self._lib = load(library)
        self.library = dict(self._lib._lpf_info)            # {"path", "build_id"}: which binary, built from which sources
        h = _P()
        rc = self._lib.lpf_create(ctypes.byref(h), int(device))
        if rc != 0:
            raise LpfError(rc, (self._lib.lpf_last_error(None) or b"").decode())
        self._h = h
        self._readers = weakref.WeakSet()
        self.device = int(device)
        self.W = self.H = 0
        self.M = 0
        self.F_masks = 0
        self.box_off = None
        self._depth = 1
        self._pin = {}                          # persistent page-locked host buffers (run_batch(pinned=True))
        # lent tensors (masks, box corners) of the runs that may still read them: in the pipelined modes a run's inputs are read
        # up to two launches after it was queued, so the references of the last few runs are kept (torch's caching allocator is
        # ordered with torch's stream, not with this context's)
        self._lent = collections.deque(maxlen=4)

    # -- plumbing ---------------------------------------------------------------------
    def _check(self, rc):
        if rc != 0:
            raise LpfError(rc, (self._lib.lpf_last_error(self._h) or b"").decode())

    def close(self):
        if getattr(self, "_h", None):
            for r in list(getattr(self, "_readers", ())):      # a reader holds pointers into its context
                r.close()
            self._lib.lpf_destroy(self._h)
            self._h = None
            for p, _, _ in self._pin.values():              # (views handed out earlier must not be used after close)
                self._lib.lpf_host_free(p)
            self._pin = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_stream(self, stream_ptr):
        """Run on the caller's HIP stream, given by its handle (``torch.cuda.current_stream().cuda_stream``; 0 is the
        null stream = torch's default stream).  ``None`` goes back to an internal stream of the context's own."""
        if stream_ptr is None:
            self._check(self._lib.lpf_use_own_stream(self._h))
        else:
            self._check(self._lib.lpf_set_stream(self._h, _P(int(stream_ptr))))

    def wait_for_stream(self, stream_ptr):
        """Device-side edge: the context's stream(s) wait for everything queued so far on ``stream_ptr`` (a HIP stream
        handle, 0 = the null stream).  Needed when the context runs on its own stream and the inputs -- or the memory of
        the output tensors -- were last touched on another stream."""
        self._check(self._lib.lpf_wait_for_stream(self._h, _P(int(stream_ptr))))

    def release_to_stream(self, stream_ptr):
        """Device-side edge the other way: ``stream_ptr`` waits for everything this context has queued."""
        self._check(self._lib.lpf_release_to_stream(self._h, _P(int(stream_ptr))))

    def sync(self):
        self._check(self._lib.lpf_sync(self._h))
        self._lent.clear()

    PIPELINED = {False: 0, 0: 0, None: 0, "off": 0, "fused": 2, 2: 2, "fused-pack": 4, 4: 4}

    def set_pipelined(self, on="fused-pack"):
        """Software-pipelined device-mode runs (lpf_set_pipelined).  ``"fused"``: the tail of a run rides in the next run's
        launch (one launch per run).  ``"fused-pack"``: the packing of lent uint8 masks rides too -- the launch of a run carries
        its mask pack, the streaming work of the run before, the tail of the one before that; a run's points and outputs are in
        use until the launch after the next.  ``False`` switches it off.  Results of a run are complete after sync() /
        release_to_stream().  Masks are set before every run; boxes set for a run travel with it (no drain)."""
        if on is True:
            raise ValueError('set_pipelined(True) meant the stream-pipelined modes 1 / 3, removed in ABI 5: use "fused" or "fused-pack"')
        if on not in self.PIPELINED:
            raise ValueError('set_pipelined: False, "fused" or "fused-pack"')
        self._check(self._lib.lpf_set_pipelined(self._h, self.PIPELINED[on]))
        self._depth = 3 if self.PIPELINED[on] else 1
        self._lent = collections.deque(self._lent, maxlen=2 * self._depth + 2)

    def set_geometry(self, mode="auto"):
        """LAB BUILDS ONLY (LPF_LIBRARY=liblpf_lab.so).  Segment / tile sizes of a run: "auto" (by launch size, what the product
        does), "small" (1024-point segments, wide tail), "small-narrow", "large" (4096) or "large-scan" (4096, prefixes from
        the scan kernel); same results."""
        if not hasattr(self._lib, "lpf_set_geometry"):
            raise LpfError(-3, "lpf_set_geometry exists in lab builds only (python -m lidar_object_detection_amd._build lab; LPF_LIBRARY=...)")
        self._check(self._lib.lpf_set_geometry(self._h, {"auto": 0, "small": 1, "large": 2, "large-scan": 3, "small-narrow": 4, "small-1024": 5}[mode]))

    ROLES = ("summaries", "box job", "lists", "box counts", "mask pack", "project+label tiles")

    def role_clock(self, reset=True):
        """LAB BUILDS ONLY.  Per role of the step launches of the software-pipelined modes since the last reset: {role: dict(span_us =
        first block start .. last block end, blocks, mean_us, longest_us)}; the first call switches the clock on.  Synchronises."""
        if not hasattr(self._lib, "lpf_lab_role_clock"):
            raise LpfError(-3, "lpf_lab_role_clock exists in lab builds only (python -m lidar_object_detection_amd._build lab; LPF_LIBRARY=...)")
        a = np.zeros((6, 5), np.uint64)
        self._check(self._lib.lpf_lab_role_clock(self._h, a.ctypes.data, 1 if reset else 0))
        out = {}
        for r, name in enumerate(self.ROLES):
            n = int(a[r, 3])
            if n:
                out[name] = dict(span_us=(int(a[r, 1]) - int(a[r, 0])) / 100.0, blocks=n, mean_us=int(a[r, 2]) / 100.0 / n, longest_us=int(a[r, 4]) / 100.0,
                                 first_start_tick=int(a[r, 0]), last_end_tick=int(a[r, 1]))
        return out

    def allreduce_metrics(self, vec, rccl_comm, op="sum"):
        """In-place all-reduce of an int64 NumPy vector over an RCCL communicator (an ncclComm_t as an integer /
        c_void_p, e.g. from ncclCommInitRank through ctypes); op: "sum", "min" or "max"."""
        a = np.ascontiguousarray(vec, dtype=np.int64)
        self._check(self._lib.lpf_allreduce_metrics(self._h, a.ctypes.data, int(a.size), {"sum": 0, "min": 1, "max": 2}[op],
                                                    rccl_comm if isinstance(rccl_comm, ctypes.c_void_p) else _P(rccl_comm)))
        return a

    def graph_begin(self):
        """Start capturing the device-mode calls made on this context into a hipGraph."""
        self._check(self._lib.lpf_graph_begin(self._h))

    def graph_end(self):
        """Finish the capture; returns an opaque handle for graph_launch()."""
        g = _P()
        self._check(self._lib.lpf_graph_end(self._h, ctypes.byref(g)))
        return g

    def graph_launch(self, g):
        self._check(self._lib.lpf_graph_launch(self._h, g))

    def graph_destroy(self, g):
        self._lib.lpf_graph_destroy(g)

    STATS = ("host_waits", "drains", "uploads", "step_launches", "box_jobs_alone", "box_jobs_riding", "blocking_uploads")

    def stats(self, reset=False):
        """dict of lpf_get_stats: what the context has done so far (host waits, drains, uploads, launches)."""
        a = np.zeros(8, np.int64)
        self._check(self._lib.lpf_get_stats(self._h, a.ctypes.data, 8, int(bool(reset))))
        return dict(zip(self.STATS, (int(v) for v in a)))

    def profile_enable(self, on=True):
        self._check(self._lib.lpf_profile_enable(self._h, int(bool(on))))

    def profile_read(self, reset=True):
        """(summed milliseconds, launches) of the event-bracketed project+label kernel."""
        ms, n = ctypes.c_double(0.0), _I64(0)
        self._check(self._lib.lpf_profile_read(self._h, ctypes.byref(ms), ctypes.byref(n), int(bool(reset))))
        return ms.value, int(n.value)

    def profile_overhead(self):
        """Milliseconds between two event records with nothing in between (what a bracket adds to a kernel)."""
        ms = ctypes.c_double(0.0)
        self._check(self._lib.lpf_profile_overhead(self._h, ctypes.byref(ms)))
        return ms.value

    # -- state ------------------------------------------------------------------------
    def set_camera(self, T_velo_to_rect, K, width, height, depth_min=0.0, depth_max=50.0):
        T = np.ascontiguousarray(T_velo_to_rect, dtype=np.float64).reshape(16)
        K3 = np.ascontiguousarray(np.asarray(K, dtype=np.float64)[:3, :3]).reshape(9)
        key = (T.tobytes(), K3.tobytes(), int(width), int(height), float(depth_min), float(depth_max))
        if key == getattr(self, "_camera", None):           # a frame loop sets the same camera every frame: nothing to do (the C call
            return                                          # would mark captured graphs stale and rebuild the box tables)
        self._check(self._lib.lpf_set_camera(self._h, T.ctypes.data, K3.ctypes.data, int(width), int(height),
                                             float(depth_min), float(depth_max)))
        self._camera = key
        self.W, self.H = int(width), int(height)

    def ensure_intrinsics(self, K, width, height):
        """The camera's K, width and height are in force -- all that lpf_prepare_boxes reads of the camera.  A frame loop that prepares
        boxes and runs frames in turn keeps the run's camera (no lpf_set_camera per frame: that call rebuilds the box tables)."""
        K3 = np.ascontiguousarray(np.asarray(K, dtype=np.float64)[:3, :3]).reshape(9)
        cam = getattr(self, "_camera", None)
        if cam is not None and cam[1] == K3.tobytes() and cam[2] == int(width) and cam[3] == int(height):
            return
        self.set_camera(np.eye(4), K3.reshape(3, 3), width, height, 0.0, 1.0)

    BINARIZE = {"astype": 0, "v3": 1, "gt0.5": 2}

    def resize_masks(self, masks):
        """cv2.resize(mask.astype(np.uint8), (W, H)) (V3:222, INTER_LINEAR) for masks [..., h, w] that are not at the camera's size:
        returns uint8 [..., H, W] -- a NumPy array for a NumPy / list input, a torch GPU tensor for a uint8 torch GPU tensor (in stream
        order).  Float masks are cast first, as the reference casts them (``astype(np.uint8)``: truncation).  Restated from OpenCV's
        C++ reference path, pinned by construction only (oracle/numpy_path.py: cv2_resize_linear_u8)."""
        if _is_torch(masks):
            import torch
            if str(masks.dtype) != "torch.uint8" or not masks.is_contiguous():
                masks = masks.to(torch.uint8).contiguous()       # (torch's float -> uint8 cast truncates, as astype does for 0 <= v < 256)
            shape = tuple(masks.shape)
            out = torch.empty(shape[:-2] + (self.H, self.W), dtype=torch.uint8, device=masks.device)
            n = int(np.prod(shape[:-2], dtype=np.int64)) if len(shape) > 2 else 1
            # Ordering (include/lpf.h, "Ordering contract"): the masks -- and the cast above, and the memory torch's caching allocator
            # just handed out for `out` -- belong to torch's current stream; the kernel runs on the context's.  An edge in, an edge
            # out: the caller may use `out` on torch's stream at once (a no-op when the context shares that stream).
            ts = torch.cuda.current_stream(masks.device).cuda_stream
            self.wait_for_stream(ts)
            self._check(self._lib.lpf_resize_masks_u8(self._h, _dev_ptr(masks) if n else None, n, shape[-2], shape[-1], _dev_ptr(out) if n else None, 1))
            self.release_to_stream(ts)
            return out
        a = np.asarray(masks)
        a = np.ascontiguousarray(a.astype(np.uint8))
        shape = a.shape
        out = np.empty(shape[:-2] + (self.H, self.W), np.uint8)
        n = int(np.prod(shape[:-2], dtype=np.int64)) if len(shape) > 2 else 1
        self._check(self._lib.lpf_resize_masks_u8(self._h, a.ctypes.data if n else None, n, shape[-2], shape[-1], out.ctypes.data if n else None, 0))
        return out

    def erode_masks(self, masks, iterations=1):
        """cv2.erode(plane, MORPH_ELLIPSE 3x3, iterations) on uint8 VALUES [..., h, w] at the planes' own size (V3:83-90, for masks that
        are eroded before they are resized): NumPy in -> NumPy out, uint8 torch GPU tensor in -> tensor out (ordered with torch's
        current stream on both sides)."""
        if _is_torch(masks):
            import torch
            if str(masks.dtype) != "torch.uint8" or not masks.is_contiguous():
                raise ValueError("device masks must be a contiguous uint8 tensor")
            shape = tuple(masks.shape)
            out = torch.empty_like(masks)
            n = int(np.prod(shape[:-2], dtype=np.int64)) if len(shape) > 2 else 1
            ts = torch.cuda.current_stream(masks.device).cuda_stream
            self.wait_for_stream(ts)
            self._check(self._lib.lpf_erode_masks_u8(self._h, _dev_ptr(masks) if n else None, n, shape[-2], shape[-1], int(iterations), _dev_ptr(out) if n else None, 1))
            self.release_to_stream(ts)
            return out
        a = np.ascontiguousarray(np.asarray(masks), dtype=np.uint8)
        out = np.empty_like(a)
        n = int(np.prod(a.shape[:-2], dtype=np.int64)) if a.ndim > 2 else 1
        self._check(self._lib.lpf_erode_masks_u8(self._h, a.ctypes.data if n else None, n, a.shape[-2], a.shape[-1], int(iterations), out.ctypes.data if n else None, 0))
        return out

    def set_mask_rects(self, rects):
        """Hint for the NEXT set_masks call: rects int32 [M,4] or [F,M,4] = (x0, y0, x1, y1), half open, pixels -- mask m of frame f is
        zero outside its rectangle (a detector's masks come cropped to their 2D boxes).  Where uint8 masks are packed as they are the
        pack then only reads what lies inside; anything else ignores the hint; results do not change as long as the word holds.  NumPy
        array (copied now) or an int32 torch GPU tensor (read when the masks are packed: keep it unchanged until then).  None clears."""
        if rects is None:
            self._check(self._lib.lpf_set_mask_rects(self._h, None, 0, 0, 0))
            return
        dev = _is_torch(rects)
        shape = tuple(rects.shape)
        if len(shape) == 2:
            shape = (1,) + shape
        if len(shape) != 3 or shape[2] != 4:
            raise ValueError("rects must be [M,4] or [F,M,4], got %s" % (shape,))
        if dev:
            if str(rects.dtype) != "torch.int32" or not rects.is_contiguous():
                raise ValueError("device rects must be a contiguous int32 tensor")
            self._lent.append(rects)
            self._check(self._lib.lpf_set_mask_rects(self._h, _dev_ptr(rects), 1, shape[0], shape[1]))
        else:
            a = np.ascontiguousarray(rects, dtype=np.int32)
            self._check(self._lib.lpf_set_mask_rects(self._h, a.ctypes.data, 0, shape[0], shape[1]))

    @staticmethod
    def mask_rects(masks):
        """Tight rectangles [..., M, 4] (x0, y0, x1, y1; half open; an empty mask gets 0, 0, 0, 0) of host masks [..., M, H, W]: what a
        detector's 2D boxes give for masks cropped to them."""
        m = np.asarray(masks) != 0
        ys, xs = m.any(axis=-1), m.any(axis=-2)
        def span(b):
            any_ = b.any(axis=-1)
            lo = np.where(any_, b.argmax(axis=-1), 0)
            hi = np.where(any_, b.shape[-1] - b[..., ::-1].argmax(axis=-1), 0)
            return lo, hi
        y0, y1 = span(ys)
        x0, x1 = span(xs)
        return np.stack([x0, y0, x1, y1], axis=-1).astype(np.int32)

    def set_masks(self, masks, erode_iters=0, v3_pipeline=False, binarize=None, lend=False):
        """masks: [M,H,W] or [F,M,H,W]; uint8/bool (nonzero = member) or float32 (reference masks).
        NumPy array, or torch tensor already on the GPU.  Float masks: ``binarize`` is "astype"
        (mask.astype(uint8) != 0, V3:222-225), "v3" (the V3:82-97 erosion block's casts; same as
        v3_pipeline=True) or "gt0.5" (mask > 0.5, Same_color.py:125 / vis.py:185).
        lend=True (GPU tensors): the tensor stays untouched until the runs that use these masks have completed, so a
        small launch may read it directly instead of packing it first, and in the "fused-pack" mode its pack rides in the
        run's launch (on_device = 2 of the C ABI).  The context keeps references to the lent tensors of the last few runs
        (until sync() at the latest); the CALLER must not rewrite a lent tensor before the run's results are complete."""
        if binarize is None:
            binarize = "v3" if v3_pipeline else "astype"
        if binarize not in self.BINARIZE:
            raise ValueError("binarize must be one of %s" % sorted(self.BINARIZE))
        dev = _is_torch(masks)
        shape = tuple(masks.shape)
        if len(shape) == 3:
            shape = (1,) + shape
        if len(shape) != 4 or (shape[1] and shape[2:] != (self.H, self.W)):
            raise ValueError("masks must be [M,%d,%d] or [F,M,%d,%d], got %s" % (self.H, self.W, self.H, self.W, shape))
        F, M = shape[0], shape[1]
        if dev:
            is_f = str(masks.dtype) == "torch.float32"
            if not is_f and str(masks.dtype) not in ("torch.uint8", "torch.bool"):
                raise ValueError("device masks must be float32, uint8 or bool")
            ptr = _dev_ptr(masks) if M else None
            keep = masks
        else:
            a = np.asarray(masks)
            is_f = a.dtype.kind == "f"
            a = np.ascontiguousarray(a, dtype=np.float32 if is_f else np.uint8)
            ptr = a.ctypes.data if M else None
            keep = a
        where = (2 if lend else 1) if dev else 0
        if is_f:
            rc = self._lib.lpf_set_masks_f32(self._h, ptr, F, M, self.BINARIZE[binarize], int(erode_iters), where)
        else:
            rc = self._lib.lpf_set_masks_u8(self._h, ptr, F, M, int(erode_iters), where)
        if dev and lend:
            self._lent.append(keep)
        del keep
        self._check(rc)
        self.F_masks, self.M = F, M

    def clear_masks(self):
        self._check(self._lib.lpf_set_masks_u8(self._h, None, 0, 0, 0, 0))
        self.F_masks, self.M = 0, 0

    def set_label_image(self, label, M):
        a = np.ascontiguousarray(label, dtype=np.uint32)
        if a.ndim == 2:
            a = a[None]
        if a.shape[1:] != (self.H, self.W):
            raise ValueError("label image must be [F,%d,%d]" % (self.H, self.W))
        self._check(self._lib.lpf_set_label_image(self._h, a.ctypes.data, a.shape[0], int(M), 0))
        self.F_masks, self.M = a.shape[0], int(M)

    def get_label_image(self):
        out = np.empty((self.F_masks, self.H, self.W), np.uint32)
        self._check(self._lib.lpf_get_label_image(self._h, out.ctypes.data, 0))
        return out

    def set_boxes(self, corners_per_frame, oriented=True):
        """corners_per_frame: list (one entry per frame) of f64 [B_f,8,3] velodyne-frame corners,
        or a single [B,8,3] array for a one-frame run."""
        if isinstance(corners_per_frame, np.ndarray):
            corners_per_frame = [corners_per_frame]
        arrs = [np.asarray(c, dtype=np.float64).reshape(-1, 8, 3) for c in corners_per_frame]
        off = np.zeros(len(arrs) + 1, np.int32)
        off[1:] = np.cumsum([a.shape[0] for a in arrs])
        cat = np.ascontiguousarray(np.concatenate(arrs, axis=0)) if arrs else np.zeros((0, 8, 3))
        self._check(self._lib.lpf_set_boxes(self._h, cat.ctypes.data if cat.size else None, off.ctypes.data,
                                            len(arrs), int(bool(oriented))))
        self.box_off = off

    def set_boxes_device(self, corners, box_off, oriented=True, lend=False):
        """Boxes from a torch float64 GPU tensor [Btot,8,3] (velodyne frame); box_off: host int32 [F+1].  No copy through the
        host and no wait (usable inside graph_begin/end with unchanged box counts).  lend=True: the tensor is read when the
        tables are built -- by the next run's launch in the pipelined modes -- and must stay unchanged until then."""
        off = np.ascontiguousarray(box_off, dtype=np.int32)
        self._check(self._lib.lpf_set_boxes_ex(self._h, _dev_ptr(corners, "float64") if int(off[-1]) else None, 2 if lend else 1, off.ctypes.data,
                                               off.shape[0] - 1, int(bool(oriented))))
        if lend:
            self._lent.append(corners)
        self.box_off = off

    def set_boxes_cam0_device(self, corners_cam0, box_off, T_cam_to_velo, filter_visible=True, oriented=True, lend=False,
                              visible=None, corners_velo=None, bbox2d=None, front=None):
        """set_boxes_cam0 with the cam-0 corners in a torch float64 GPU tensor [Btot,8,3] (and optional GPU output tensors:
        visible uint8 [Btot], corners_velo float64 [Btot,8,3], bbox2d float64 [Btot,4], front int32 [Btot]).  The reference's
        per-frame box preparation (V3:556-562) entirely on the device, without a wait; in the pipelined modes it rides in the
        next run's launch.  lend as in set_boxes_device."""
        off = np.ascontiguousarray(box_off, dtype=np.int32)
        T = np.ascontiguousarray(T_cam_to_velo, dtype=np.float64).reshape(16)
        self._check(self._lib.lpf_set_boxes_cam0(self._h, _dev_ptr(corners_cam0, "float64") if int(off[-1]) else None, 2 if lend else 1,
                                                 off.ctypes.data, off.shape[0] - 1, T.ctypes.data, int(bool(filter_visible)), int(bool(oriented)),
                                                 _dev_ptr(visible, "uint8"), _dev_ptr(corners_velo, "float64"), _dev_ptr(bbox2d, "float64"),
                                                 _dev_ptr(front, "int32")))
        if lend:
            self._lent.append(corners_cam0)
        self.box_off = off

    def set_boxes_cam0(self, corners_cam0_per_frame, T_cam_to_velo, filter_visible=True, oriented=True, want_outputs=True):
        """The reference's per-frame box preparation (filter_visible_bboxes + transform_bboxes_to_velodyne, V3:556-562) and
        set_boxes in one device-side step.  corners_cam0_per_frame: list of f64 [B_f,8,3] (or one array).  Box indices of later
        results refer to the GIVEN boxes; dropped ones have zero counts.  Returns per frame (visible bool[B_f], corners_velo
        [B_f,8,3], bbox2d [B_f,4], front int32[B_f]) unless want_outputs is False."""
        if isinstance(corners_cam0_per_frame, np.ndarray):
            corners_cam0_per_frame = [corners_cam0_per_frame]
        arrs = [np.asarray(c, dtype=np.float64).reshape(-1, 8, 3) for c in corners_cam0_per_frame]
        off = np.zeros(len(arrs) + 1, np.int32)
        off[1:] = np.cumsum([a.shape[0] for a in arrs])
        cat = np.ascontiguousarray(np.concatenate(arrs, axis=0)) if arrs else np.zeros((0, 8, 3))
        T = np.ascontiguousarray(T_cam_to_velo, dtype=np.float64).reshape(16)
        B = int(off[-1])
        vis, cv = np.zeros(B, np.uint8), np.zeros((B, 8, 3), np.float64)
        bb, fr = np.zeros((B, 4), np.float64), np.zeros(B, np.int32)
        w = want_outputs and B > 0
        self._check(self._lib.lpf_set_boxes_cam0(self._h, cat.ctypes.data if B else None, 0, off.ctypes.data, len(arrs), T.ctypes.data,
                                                 int(bool(filter_visible)), int(bool(oriented)), vis.ctypes.data if w else None,
                                                 cv.ctypes.data if w else None, bb.ctypes.data if w else None, fr.ctypes.data if w else None))
        self.box_off = off
        if not want_outputs:
            return None
        return [(vis[a:b].astype(bool), cv[a:b], bb[a:b], fr[a:b]) for a, b in zip(off[:-1], off[1:])]

    def clear_boxes(self):
        self._check(self._lib.lpf_set_boxes(self._h, None, None, 0, 1))
        self.box_off = None

    def points_in_boxes(self, points, corners, oriented=True):
        """bool [B,k]: row b is the reference's oriented_point_in_bbox(points, corners[b]) (or point_in_bbox)."""
        p = np.ascontiguousarray(points, dtype=np.float32)
        if p.ndim != 2 or p.shape[1] not in (3, 4):
            raise ValueError("points must be [k,3] or [k,4]")
        c = np.ascontiguousarray(corners, dtype=np.float64).reshape(-1, 8, 3)
        k, B = p.shape[0], c.shape[0]
        out = np.zeros((B, k), np.uint8)
        if k and B:
            self._check(self._lib.lpf_points_in_boxes(self._h, p.ctypes.data, k, p.shape[1], c.ctypes.data, B,
                                                      int(bool(oriented)), out.ctypes.data, 0))
        return out.astype(bool)

    def depth_image(self, points):
        """(D f64[H,W], winner int32[H,W]) of f32[N,4] host points: depth of the last valid point per pixel."""
        p = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 4)
        D, win = np.empty((self.H, self.W), np.float64), np.empty((self.H, self.W), np.int32)
        self._check(self._lib.lpf_depth_image(self._h, p.ctypes.data if len(p) else None, p.shape[0], 0,
                                              D.ctypes.data, win.ctypes.data))
        return D, win

    def prepare_boxes(self, corners_cam0, T_cam_to_velo):
        """(visible bool[B], corners_velo f64[B,8,3], bbox2d f64[B,4], front int32[B]) of B annotated boxes
        given by their cam-0 corners: filter_visible_bboxes + transform_bboxes_to_velodyne + V4's projected
        2D box, computed on the GPU with the reference's arithmetic (needs set_camera first)."""
        c = np.ascontiguousarray(corners_cam0, dtype=np.float64).reshape(-1, 8, 3)
        T = np.ascontiguousarray(T_cam_to_velo, dtype=np.float64).reshape(16)
        B = c.shape[0]
        vis, cv = np.zeros(B, np.uint8), np.zeros((B, 8, 3), np.float64)
        bb, fr = np.zeros((B, 4), np.float64), np.zeros(B, np.int32)
        if B:
            self._check(self._lib.lpf_prepare_boxes(self._h, c.ctypes.data, B, T.ctypes.data, vis.ctypes.data, cv.ctypes.data,
                                                    bb.ctypes.data, fr.ctypes.data))
        return vis.astype(bool), cv, bb, fr

    # -- the hot path, host arrays ------------------------------------------------------
    def run(self, points, **kw):
        """One frame of f32[N,4] host points -> dict of NumPy results (see run_batch)."""
        return self.run_batch([points], **kw)[0]

    def _pinned(self, name, shape, dtype):
        """A persistent page-locked host array of the context (grow-only; lpf_host_alloc = hipHostMalloc, no torch involved): copies
        from the GPU into it are DMA transfers the call does not stage through pageable memory, and a frame loop does not allocate
        (and page in) megabytes per call."""
        need = int(np.prod(shape)) * np.dtype(dtype).itemsize
        ent = self._pin.get(name)
        if ent is None or ent[1] < need:
            if ent is not None:
                self._lib.lpf_host_free(ent[0])
                del self._pin[name]
            cap = max(need + need // 4, 4096)
            p = self._lib.lpf_host_alloc(cap)
            if not p:
                raise LpfError(-4, (self._lib.lpf_last_error(None) or b"lpf_host_alloc failed").decode())
            ent = self._pin[name] = (p, cap, np.frombuffer((ctypes.c_uint8 * cap).from_address(p), dtype=np.uint8))
        return ent[2][:need].view(dtype).reshape(shape)

    def run_batch(self, frames, want_uv=True, want_label=True, want_float=False, want_lists=True,
                  inst_cap=None, want_valid_uv=False, pinned=False):
        """frames: list of f32[N_f,4] arrays.  Returns one dict per frame with
        u, v (int32), label_bits, valid_idx, inst_lists, inst_count, count_mb, best_box, best_cnt,
        n_valid, n_labelled (+ depth, uf, vf with want_float; + u_valid, v_valid, label_valid with
        want_valid_uv: the values at the valid points only -- with want_uv/want_label off, a quarter of the read-back).
        pinned=True: the result arrays are views into page-locked buffers the context owns and reuses -- valid until the next
        run on this context (copy what must live longer); the frame loops use it."""
        scan = frames[0] if (len(frames) == 1 and isinstance(frames[0], Scan)) else None
        dev_pts = frames[0] if (len(frames) == 1 and _is_torch(frames[0])) else None
        if scan is not None:                     # points already in HBM (ScanReader): no host staging
            frames = [scan.points]
        elif dev_pts is not None:                # ... or a float32 [N,4] torch tensor on the GPU (ordered with torch's current stream)
            if dev_pts.ndim != 2 or dev_pts.shape[1] != 4:
                raise ValueError("device points must be a float32 tensor [N,4]")
            frames = [np.empty((int(dev_pts.shape[0]), 4), np.float32)]      # (only its shape is used below)
        elif any(isinstance(p, Scan) or _is_torch(p) for p in frames):
            raise ValueError("a Scan from a ScanReader / a GPU tensor of points is processed on its own (one frame per run)")
        frames = [np.ascontiguousarray(p, dtype=np.float32).reshape(-1, 4) for p in frames]
        F = len(frames)
        off = np.zeros(F + 1, np.int64)
        off[1:] = np.cumsum([p.shape[0] for p in frames])
        n = int(off[-1])
        batch_dev = None
        if F > 1 and n and scan is None and dev_pts is None:
            # several host frames: each goes to its place in ONE device tensor -- no concatenation of the batch on the host first
            # (20 real frames are 37 MB: the copy cost as much as their kernels a hundred times over)
            try:
                import torch
                if torch.cuda.is_available():
                    batch_dev = torch.empty((n, 4), dtype=torch.float32, device=torch.device("cuda", self.device))
                    for f_, p_ in enumerate(frames):
                        if p_.shape[0]:
                            batch_dev[int(off[f_]):int(off[f_ + 1])].copy_(torch.from_numpy(p_))
                    self.wait_for_stream(torch.cuda.current_stream(batch_dev.device).cuda_stream)
            except ImportError:
                batch_dev = None
        if batch_dev is not None:
            pts, pts_ptr, pts_dev = None, batch_dev.data_ptr(), 1
        else:
            pts = np.concatenate(frames, axis=0) if F > 1 else frames[0]
            pts_ptr, pts_dev = (pts.ctypes.data if n else None), 0
        if scan is not None:
            scan._check_live()
            pts_ptr, pts_dev = (scan.dev_ptr if n else None), 1
        elif dev_pts is not None:
            import torch
            self.wait_for_stream(torch.cuda.current_stream(dev_pts.device).cuda_stream)      # the tensor was produced on torch's stream
            pts_ptr, pts_dev = (_dev_ptr(dev_pts, "float32") if n else None), 1
        M = self.M if self.F_masks else 0
        Btot = int(self.box_off[-1]) if self.box_off is not None else 0
        if inst_cap is None:
            inst_cap = max(int(max(p.shape[0] for p in frames)), 1)
        new = (lambda name, shape, dt: self._pinned(name, shape, dt)) if pinned else (lambda name, shape, dt: np.empty(shape, dt))
        while True:
            o = Outputs()
            o.on_device = 0
            uv = new("uv", (n, 2), np.int32) if want_uv else None
            lab = new("lab", (n,), np.uint32) if want_label else None
            dep = new("dep", (n,), np.float64) if want_float else None
            uf = new("uf", (n,), np.float64) if want_float else None
            vf = new("vf", (n,), np.float64) if want_float else None
            vidx = new("vidx", (n,), np.int64) if want_lists else None
            uvv = new("uvv", (n, 2), np.int32) if (want_valid_uv and want_lists) else None     # only the first n_valid rows come back
            labv = new("labv", (n,), np.uint32) if (want_valid_uv and want_lists) else None
            iidx = new("iidx", (F, inst_cap), np.int64) if (want_lists and M) else None
            cmb = new("cmb", (max(M * Btot, 1),), np.int32)
            cmb[:] = 0
            summ = new("summ", (F,), SUMMARY_DTYPE)
            summ.view(np.uint8)[:] = 0
            for name, arr in (("uv", uv), ("label_bits", lab), ("depth", dep), ("u_f", uf), ("v_f", vf),
                              ("valid_idx", vidx), ("inst_idx", iidx), ("count_mb", cmb), ("summary", summ),
                              ("uv_valid", uvv), ("label_valid", labv)):
                setattr(o, name, arr.ctypes.data if arr is not None else None)
            o.inst_cap = inst_cap
            self._check(self._lib.lpf_run_batch(self._h, pts_ptr, off.ctypes.data, F, pts_dev, ctypes.byref(o)))
            if iidx is not None and summ["inst_overflow"].any():
                inst_cap = int(summ["inst_off"][:, 32].max())      # exact size now known: run again
                continue
            break
        res = []
        for f in range(F):
            a, b = int(off[f]), int(off[f + 1])
            s = summ[f]
            r = dict(n_valid=int(s["n_valid"]), n_labelled=int(s["n_labelled"]),
                     inst_count=s["inst_count"][:M].copy(), best_box=s["best_box"][:M].copy(),
                     best_cnt=s["best_cnt"][:M].copy())
            if want_uv:
                r["u"], r["v"] = uv[a:b, 0], uv[a:b, 1]
            if want_label:
                r["label_bits"] = lab[a:b]
            if want_float:
                r["depth"], r["uf"], r["vf"] = dep[a:b], uf[a:b], vf[a:b]
            if want_lists:
                r["valid_idx"] = vidx[a:a + r["n_valid"]]
                if uvv is not None:
                    r["uv_valid"] = uvv[a:a + r["n_valid"]]                      # contiguous [n_valid, 2]; u / v are its columns
                    r["u_valid"], r["v_valid"] = r["uv_valid"][:, 0], r["uv_valid"][:, 1]
                    r["label_valid"] = labv[a:a + r["n_valid"]]
                r["inst_lists"] = [iidx[f, int(s["inst_off"][m]):int(s["inst_off"][m + 1])] for m in range(M)] \
                    if iidx is not None else []
            if self.box_off is not None:
                b0, b1 = int(self.box_off[f]), int(self.box_off[f + 1])
                r["count_mb"] = cmb[M * b0:M * b1].reshape(M, b1 - b0).astype(np.int64)
            else:
                r["count_mb"] = np.zeros((M, 0), np.int64)
            res.append(r)
        return res

    # -- the hot path, device tensors (asynchronous) -------------------------------------
    def run_device(self, pts, frame_off, uv=None, label_bits=None, depth=None, u_f=None, v_f=None,
                   valid_idx=None, inst_idx=None, inst_cap=0, count_mb=None, summary=None, uv_valid=None, label_valid=None):
        """Enqueue one batch on the context's stream.  pts: torch float32 [Ntot,4] on the GPU;
        outputs: preallocated torch tensors (None = not wanted); summary: uint8 [F*928].
        frame_off: int64 NumPy/sequence [F+1] (host)."""
        off = np.ascontiguousarray(frame_off, dtype=np.int64)
        F = off.shape[0] - 1
        o = Outputs()
        o.on_device = 1
        o.uv = _dev_ptr(uv, "int32")
        o.label_bits = _dev_ptr(label_bits)
        o.depth, o.u_f, o.v_f = _dev_ptr(depth, "float64"), _dev_ptr(u_f, "float64"), _dev_ptr(v_f, "float64")
        o.valid_idx = _dev_ptr(valid_idx, "int64")
        o.inst_idx = _dev_ptr(inst_idx, "int64")
        o.inst_cap = int(inst_cap)
        o.count_mb = _dev_ptr(count_mb, "int32")
        o.summary = _dev_ptr(summary)
        o.uv_valid, o.label_valid = _dev_ptr(uv_valid, "int32"), _dev_ptr(label_valid)
        self._check(self._lib.lpf_run_batch(self._h, _dev_ptr(pts, "float32"), off.ctypes.data, F, 1,
                                            ctypes.byref(o)))

    def make_frame_step(self, pts, masks_u8=None, mask_rects=None, boxes_cam0=None, T_cam_to_velo=None, filter_visible=True, oriented=True, **outs):
        """Pre-marshal ONE frame of a stream -- scan, lent uint8 masks [M,H,W] (+ their rectangles [M,4]), lent cam-0 box corners
        [B,8,3], output tensors as in make_device_step -- into an lpf_frame_job and return a zero-argument callable that makes the one
        C call (lpf_run_frame): masks, rectangles, boxes and the run in a single FFI crossing.  Lent tensors stay unchanged until the
        frame's results are complete (pipelined modes: two launches later)."""
        j = FrameJob()
        n = int(pts.shape[0])
        j.pts, j.n_points = _dev_ptr(pts, "float32"), n
        keep = [pts, outs]
        if masks_u8 is not None:
            shape = tuple(masks_u8.shape)
            if len(shape) != 3 or shape[1:] != (self.H, self.W) or str(masks_u8.dtype) != "torch.uint8":
                raise ValueError("masks_u8 must be a torch.uint8 GPU tensor [M,H,W]")
            j.masks, j.n_masks = _dev_ptr(masks_u8), shape[0]
            self.F_masks, self.M = 1, shape[0]
            keep.append(masks_u8)
            if mask_rects is not None:
                if tuple(mask_rects.shape) != (shape[0], 4) or str(mask_rects.dtype) != "torch.int32" or not mask_rects.is_contiguous():
                    raise ValueError("mask_rects must be a contiguous torch.int32 GPU tensor [M,4]")
                j.mask_rects = _dev_ptr(mask_rects)
                keep.append(mask_rects)
        if boxes_cam0 is not None:
            Tcv = np.ascontiguousarray(T_cam_to_velo, dtype=np.float64).reshape(16)
            j.corners_cam0, j.n_boxes, j.T_cam_to_velo = _dev_ptr(boxes_cam0, "float64") if boxes_cam0.shape[0] else None, int(boxes_cam0.shape[0]), Tcv.ctypes.data
            j.filter_visible, j.oriented = int(bool(filter_visible)), int(bool(oriented))
            self.box_off = np.array([0, int(boxes_cam0.shape[0])], np.int32)
            keep += [boxes_cam0, Tcv]
        o = j.out
        o.on_device = 1
        o.uv = _dev_ptr(outs.get("uv"), "int32")
        o.label_bits = _dev_ptr(outs.get("label_bits"))
        o.depth, o.u_f, o.v_f = (_dev_ptr(outs.get("depth"), "float64"), _dev_ptr(outs.get("u_f"), "float64"), _dev_ptr(outs.get("v_f"), "float64"))
        o.valid_idx = _dev_ptr(outs.get("valid_idx"), "int64")
        o.inst_idx = _dev_ptr(outs.get("inst_idx"), "int64")
        o.inst_cap = int(outs.get("inst_cap", 0))
        o.count_mb = _dev_ptr(outs.get("count_mb"), "int32")
        o.summary = _dev_ptr(outs.get("summary"))
        o.uv_valid, o.label_valid = _dev_ptr(outs.get("uv_valid"), "int32"), _dev_ptr(outs.get("label_valid"))
        run, h, check, ref = self._lib.lpf_run_frame, self._h, self._check, ctypes.byref(j)

        def fn(_keep=(j, keep)):
            rc = run(h, ref)
            if rc:
                check(rc)
        return fn

    def make_device_step(self, pts, frame_off, masks_u8=None, erode_iters=0, lend=False, boxes_cam0=None, box_off=None,
                         T_cam_to_velo=None, filter_visible=True, oriented=True, mask_rects=None, **outs):
        """Pre-marshal one device-mode step -- optional u8 masks, optional per-step boxes (the reference's per-frame box
        preparation from cam-0 corners, V3:556-562), run_batch -- and return a zero-argument callable that only performs the C
        calls: for launch-bound loops.
        lend=False: the mask tensor may be rewritten, in stream order, right after the step call returns (it is packed by a
        launch of its own at the call).  lend=True passes masks (and box corners) as LENT (on_device = 2): nothing is copied or
        packed at the call -- in the "fused-pack" mode the pack and the box set-up ride in the step's launch, small sparse launches
        read the masks directly -- but the tensors must then stay UNCHANGED until the step's results are complete (in the
        pipelined modes a step's inputs are read up to two launches later).
        mask_rects: int32 torch GPU tensor [F,M,4], the masks' rectangles (set_mask_rects), lent like the masks."""
        off = np.ascontiguousarray(frame_off, dtype=np.int64)
        F = off.shape[0] - 1
        o = Outputs()
        o.on_device = 1
        o.uv = _dev_ptr(outs.get("uv"), "int32")
        o.label_bits = _dev_ptr(outs.get("label_bits"))
        o.depth, o.u_f, o.v_f = (_dev_ptr(outs.get("depth"), "float64"), _dev_ptr(outs.get("u_f"), "float64"),
                                 _dev_ptr(outs.get("v_f"), "float64"))
        o.valid_idx = _dev_ptr(outs.get("valid_idx"), "int64")
        o.inst_idx = _dev_ptr(outs.get("inst_idx"), "int64")
        o.inst_cap = int(outs.get("inst_cap", 0))
        o.count_mb = _dev_ptr(outs.get("count_mb"), "int32")
        o.summary = _dev_ptr(outs.get("summary"))
        o.uv_valid, o.label_valid = _dev_ptr(outs.get("uv_valid"), "int32"), _dev_ptr(outs.get("label_valid"))
        lib, h, check = self._lib, self._h, self._check
        p_pts, p_off, p_out = _P(_dev_ptr(pts, "float32")), _P(off.ctypes.data), ctypes.byref(o)
        run, setm, setb = lib.lpf_run_batch, lib.lpf_set_masks_u8, lib.lpf_set_boxes_cam0
        where = 2 if lend else 1
        calls = []
        keep = [off, o, pts, outs]
        if masks_u8 is not None:
            shape = tuple(masks_u8.shape)
            if len(shape) == 3:
                shape = (1,) + shape
            if shape[0] != F or shape[2:] != (self.H, self.W) or str(masks_u8.dtype) != "torch.uint8":
                raise ValueError("masks_u8 must be a torch.uint8 GPU tensor [F,M,H,W]")
            p_m, M, it = _P(_dev_ptr(masks_u8)), shape[1], int(erode_iters)
            self.F_masks, self.M = F, M
            keep.append(masks_u8)
            if mask_rects is not None:
                if tuple(mask_rects.shape) != (F, M, 4) or str(mask_rects.dtype) != "torch.int32" or not mask_rects.is_contiguous():
                    raise ValueError("mask_rects must be a contiguous torch.int32 GPU tensor [F,M,4]")
                p_r, setr = _P(_dev_ptr(mask_rects)), lib.lpf_set_mask_rects
                keep.append(mask_rects)
                calls.append(lambda: setr(h, p_r, 1, F, M))
            calls.append(lambda: setm(h, p_m, F, M, it, where))
        if boxes_cam0 is not None:
            boff = np.ascontiguousarray(box_off, dtype=np.int32)
            if boff.shape[0] != F + 1:
                raise ValueError("box_off must have F + 1 entries")
            Tcv = np.ascontiguousarray(T_cam_to_velo, dtype=np.float64).reshape(16)
            p_b = _P(_dev_ptr(boxes_cam0, "float64")) if int(boff[-1]) else None
            p_bo, p_T, fv, ori = _P(boff.ctypes.data), _P(Tcv.ctypes.data), int(bool(filter_visible)), int(bool(oriented))
            self.box_off = boff
            keep += [boxes_cam0, boff, Tcv]
            calls.append(lambda: setb(h, p_b, where, p_bo, F, p_T, fv, ori, None, None, None, None))
        calls.append(lambda: run(h, p_pts, p_off, F, 1, p_out))
        calls = tuple(calls)

        def fn(_keep=tuple(keep)):
            for call in calls:
                rc = call()
                if rc:
                    check(rc)
        return fn
