"""Host callers with page-locked result buffers (LpfContext.run_batch(pinned=True), what run_frames uses): the compact results are
written to host memory by a kernel that reads their lengths on the device (lpf_results_to_host) -- one launch, one host wait -- and
must be, byte for byte, what the copy-engine path (pageable buffers: summaries first, then the filled part of every list) returns."""
import numpy as np
import pytest

from oracle import cpu_oracle as orc

pytestmark = pytest.mark.gpu

KEYS = ("n_valid", "n_labelled", "inst_count", "best_box", "best_cnt", "valid_idx", "u_valid", "v_valid", "label_valid", "count_mb")


def _same(a, b, why):
    assert len(a) == len(b)
    for f, (x, y) in enumerate(zip(a, b)):
        for k in KEYS:
            if k in x or k in y:
                assert np.array_equal(x[k], y[k]), why + (f, k)
        assert len(x.get("inst_lists", [])) == len(y.get("inst_lists", [])), why + (f,)
        for m, (p, q) in enumerate(zip(x.get("inst_lists", []), y.get("inst_lists", []))):
            assert np.array_equal(p, q), why + (f, m)


@pytest.mark.parametrize("mode", [False, "fused-pack"])
def test_pinned_results_equal_the_copy_path(calib, mode):
    from lidar_object_detection_amd import synthetic as S
    from lidar_object_detection_amd._native import LpfContext
    _, T, K, W, H = S.default_calibration(calib)
    scs = [S.scene(n, n_masks=5, n_boxes=7, seed=4100 + i, calib=calib) for i, n in enumerate((90_000, 1, 130_001, 64, 4097))]
    behind = scs[0]["points"].copy(); behind[:, 0] = -np.abs(behind[:, 0]) - 1.0          # a frame with no valid point at all
    clouds = [scs[0]["points"], scs[1]["points"], np.zeros((0, 4), np.float32), scs[2]["points"], behind[:5000], scs[3]["points"], scs[4]["points"]]
    masks = np.stack([scs[i % 5]["masks"] for i in range(len(clouds))])
    boxes = [scs[i % 5]["corners_velo"][:(i % 4) * 2] for i in range(len(clouds))]         # 0, 2, 4, 6 boxes: ragged, one frame without
    with LpfContext(0) as ctx:
        ctx.set_pipelined(mode)
        ctx.set_camera(T, K, W, H, 0.0, 45.0)
        for batch in (list(range(len(clouds))), [0], [2], [4, 2], [3, 0, 6]):
            pts = [clouds[i] for i in batch]
            ctx.set_masks(masks[batch])
            ctx.set_boxes([boxes[i] for i in batch])
            for kw in (dict(want_uv=False, want_label=False, want_valid_uv=True), dict(want_valid_uv=True, want_float=True), dict()):
                ctx.run_batch(pts, pinned=True, **kw)                      # (buffers grow to size)
                ctx.stats(reset=True)
                got = ctx.run_batch(pts, pinned=True, **kw)
                st_pin = ctx.stats(reset=True)
                got = [{k: (v.copy() if isinstance(v, np.ndarray) else [l.copy() for l in v] if isinstance(v, list) else v) for k, v in r.items()} for r in got]
                want = ctx.run_batch(pts, pinned=False, **kw)
                st_cpy = ctx.stats(reset=True)
                _same(got, want, (mode, tuple(batch), tuple(sorted(kw))))
                for k in ("u", "v", "label_bits", "depth"):
                    for x, y in zip(got, want):
                        assert (k in x) == (k in y) and (k not in x or np.array_equal(x[k], y[k], equal_nan=True)), (k, batch)
                assert st_pin["host_waits"] < st_cpy["host_waits"], (st_pin, st_cpy)     # one wait instead of two
            # ... and both are the oracle's
            for i, r in zip(batch, want):
                ref = orc.run(clouds[i], T, K, W, H, 0.0, 45.0, label_img=orc.pack_masks(masks[i], 0, H, W), M=5, corners=boxes[i], want_float=False)
                assert r["n_valid"] == ref["n_valid"] and np.array_equal(r["valid_idx"], ref["valid_idx"])
                assert np.array_equal(r["count_mb"], ref["count_mb"]) and all(np.array_equal(a, b) for a, b in zip(r["inst_lists"], ref["inst_lists"]))
        # lists longer than the capacity given: the summaries say so in either path, and the second run has room
        ctx.set_masks(masks[:1]); ctx.set_boxes([boxes[1]])
        a = ctx.run_batch([clouds[0]], pinned=True, inst_cap=8, want_valid_uv=True)
        a = [{k: (v.copy() if isinstance(v, np.ndarray) else [l.copy() for l in v] if isinstance(v, list) else v) for k, v in r.items()} for r in a]
        b = ctx.run_batch([clouds[0]], pinned=False, inst_cap=8, want_valid_uv=True)
        _same(a, b, ("overflow",))
        assert sum(len(l) for l in a[0]["inst_lists"]) > 8
        # no masks at all: no lists, nothing to copy for them
        ctx.set_masks(np.zeros((1, 0, H, W), np.uint8))
        a = ctx.run_batch([clouds[3]], pinned=True, want_valid_uv=True)
        a = [{k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in r.items()} for r in a]
        b = ctx.run_batch([clouds[3]], pinned=False, want_valid_uv=True)
        _same(a, b, ("no masks",))


def test_result_buffers_inside_one_page_locked_arena(calib):
    """A C host that carves its result buffers out of ONE page-locked allocation (pointers into the middle of it): whichever way the
    library takes for them, every buffer holds exactly its own results and not a byte outside it is touched."""
    import ctypes
    from lidar_object_detection_amd import synthetic as S
    from lidar_object_detection_amd._native import LpfContext, Outputs, SUMMARY_DTYPE
    _, T, K, W, H = S.default_calibration(calib)
    sc = S.scene(70_000, n_masks=4, n_boxes=5, seed=777, calib=calib)
    n, M, B = len(sc["points"]), 4, 5
    with LpfContext(0) as ctx:
        ctx.set_camera(T, K, W, H, 0.0, 45.0)
        ctx.set_masks(sc["masks"][None])
        ctx.set_boxes([sc["corners_velo"]])
        want = ctx.run_batch([sc["points"]], want_uv=False, want_label=False, want_valid_uv=True)[0]
        lib = ctx._lib
        sizes = dict(summary=SUMMARY_DTYPE.itemsize, valid_idx=8 * n, uv_valid=8 * n, label_valid=4 * n, inst_idx=8 * n, count_mb=4 * M * B)
        gap = 4096
        total = gap + sum(v + gap for v in sizes.values())
        base = lib.lpf_host_alloc(total)
        assert base
        try:
            arena = np.frombuffer((ctypes.c_uint8 * total).from_address(base), dtype=np.uint8)
            arena[:] = 0xA5
            off, at = {}, gap
            for k, v in sizes.items():
                off[k] = at
                at += v + gap
            o = Outputs()
            o.on_device = 0
            for k in sizes:
                setattr(o, k, base + off[k])
            o.inst_cap = n
            pts = np.ascontiguousarray(sc["points"], dtype=np.float32)
            foff = np.array([0, n], np.int64)
            ctx._check(lib.lpf_run_batch(ctx._h, pts.ctypes.data, foff.ctypes.data, 1, 0, ctypes.byref(o)))
            view = lambda k, dt: arena[off[k]:off[k] + sizes[k]].view(dt)                      # noqa: E731
            sm = view("summary", SUMMARY_DTYPE)[0]
            nv, tot = int(sm["n_valid"]), int(sm["inst_off"][32])
            assert nv == want["n_valid"] and np.array_equal(view("valid_idx", np.int64)[:nv], want["valid_idx"])
            assert np.array_equal(view("uv_valid", np.int32).reshape(-1, 2)[:nv], want["uv_valid"])
            assert np.array_equal(view("label_valid", np.uint32)[:nv], want["label_valid"])
            assert np.array_equal(view("inst_idx", np.int64)[:tot], np.concatenate(want["inst_lists"]))
            assert np.array_equal(view("count_mb", np.int32).reshape(M, B), want["count_mb"])
            # nothing beyond the filled parts, nothing between the buffers
            assert (view("valid_idx", np.uint8)[8 * nv:] == 0xA5).all() and (view("inst_idx", np.uint8)[8 * tot:] == 0xA5).all()
            guard = np.ones(total, bool)
            for k, v in sizes.items():
                guard[off[k]:off[k] + v] = False
            assert (arena[guard] == 0xA5).all()
        finally:
            lib.lpf_host_free(base)
