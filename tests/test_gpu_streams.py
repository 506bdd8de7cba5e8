"""Stream-ordering contract of device mode (include/lpf.h, "Ordering contract") and staleness of captured graphs.

Round 1's bench once died inside torch's set-up gather (gpurun_out/bench_s1.*: HSA_STATUS_ERROR_EXCEPTION 0x1016 =
a device-side abort() of torch's index bounds check): `set_stream(0)` silently detached the context from torch's
default stream, so its warm-up kernels wrote into an output tensor whose memory the caching allocator had recycled
from an index tensor that a still-queued gather was going to read.  These tests pin the fixed behaviour: handle 0 is
honoured, and a context on its own stream gets explicit device-side edges -- both without any device-wide sync.
"""
import numpy as np
import pytest

from oracle import cpu_oracle as orc

pytestmark = pytest.mark.gpu


def _delayed_inputs(torch, dev, sc):
    """Points and masks that only become correct at the END of a long chain of kernels on the current stream: a consumer
    that is not ordered behind that stream sees zeros."""
    a = torch.randn(4096, 4096, device=dev)
    for _ in range(40):                                     # tens of milliseconds of queued work
        a = (a @ a).clamp_(-1.0, 1.0)
    gate = (a.sum() * 0.0).to(torch.float32)                # 0.0, available only when the chain is done
    pts = torch.zeros((len(sc["points"]), 4), dtype=torch.float32, device=dev)
    pts.add_(torch.from_numpy(sc["points"]).to(dev, non_blocking=True) + gate)
    masks = torch.zeros(sc["masks"].shape, dtype=torch.uint8, device=dev)
    masks.add_(torch.from_numpy(sc["masks"]).to(dev, non_blocking=True) + gate.to(torch.uint8))
    return pts, masks


def _outputs(torch, dev, n, M, Bx, summary_bytes):
    return dict(uv=torch.empty((n, 2), dtype=torch.int32, device=dev), label_bits=torch.empty(n, dtype=torch.int32, device=dev),
                valid_idx=torch.empty(n, dtype=torch.int64, device=dev), inst_idx=torch.empty(n, dtype=torch.int64, device=dev),
                count_mb=torch.zeros(M * Bx, dtype=torch.int32, device=dev),
                summary=torch.zeros(summary_bytes, dtype=torch.uint8, device=dev))


def _check(o, sc, T, K, W, H, M, Bx, summary_dtype):
    # .cpu() runs on torch's current stream: it is ordered behind the context's work by the stream / the release edge
    sm = np.frombuffer(o["summary"].cpu().numpy().tobytes(), summary_dtype)[0]
    ref = orc.run(sc["points"], T, K, W, H, 0.0, 30.0, label_img=orc.pack_masks(sc["masks"], 0, H, W), M=M,
                  corners=sc["corners_velo"], want_float=False)
    uv = o["uv"].cpu().numpy()
    assert np.array_equal(uv[:, 0], ref["u"]) and np.array_equal(uv[:, 1], ref["v"])
    assert np.array_equal(o["label_bits"].cpu().numpy().view(np.uint32), ref["label_bits"])
    assert int(sm["n_valid"]) == ref["n_valid"] and ref["n_valid"] > 1000
    assert np.array_equal(o["valid_idx"].cpu().numpy()[:ref["n_valid"]], ref["valid_idx"])
    assert np.array_equal(sm["inst_count"][:M], ref["inst_count"]) and int(ref["inst_count"].sum()) > 0
    assert np.array_equal(o["count_mb"].cpu().numpy().reshape(M, Bx), ref["count_mb"])


@pytest.mark.parametrize("how", ["shared_default_stream", "own_stream_with_edges", "own_stream_pipelined_with_edges"])
def test_device_mode_is_ordered_without_a_device_sync(calib, how):
    import torch
    from lidar_object_detection_amd import synthetic as S
    from lidar_object_detection_amd._native import LpfContext, SUMMARY_DTYPE
    _, T, K, W, H = S.default_calibration(calib)
    dev = torch.device("cuda", 0)
    n, M, Bx = 200_000, 6, 9
    sc = S.scene(n, n_masks=M, n_boxes=Bx, seed=77)
    cur = torch.cuda.current_stream(dev)
    assert cur.cuda_stream == 0, "this test wants torch's default (null) stream"
    with LpfContext(0) as ctx:
        if how == "shared_default_stream":
            ctx.set_stream(cur.cuda_stream)                 # handle 0: the null stream itself
        elif how == "own_stream_pipelined_with_edges":
            ctx.set_pipelined("fused-pack")
        ctx.set_camera(T, K, W, H, 0.0, 30.0)
        ctx.set_boxes(sc["corners_velo"])
        # warm run on other data: scratch allocation / table uploads (which synchronise) happen here, not below
        warm = S.scene(n, n_masks=M, n_boxes=Bx, seed=78)
        ow = _outputs(torch, dev, n, M, Bx, SUMMARY_DTYPE.itemsize)
        ctx.set_masks(torch.from_numpy(warm["masks"]).to(dev))
        ctx.run_device(torch.from_numpy(warm["points"]).to(dev), np.array([0, n], np.int64), inst_cap=n, **ow)
        ctx.sync()
        torch.cuda.synchronize(dev)
        # ---- from here on: no host-side wait until the results are read ----
        pts, masks = _delayed_inputs(torch, dev, sc)
        o = _outputs(torch, dev, n, M, Bx, SUMMARY_DTYPE.itemsize)
        if how != "shared_default_stream":
            ctx.wait_for_stream(cur.cuda_stream)
        ctx.set_masks(masks)
        ctx.run_device(pts, np.array([0, n], np.int64), inst_cap=n, **o)
        if how != "shared_default_stream":
            ctx.release_to_stream(cur.cuda_stream)
        _check(o, sc, T, K, W, H, M, Bx, SUMMARY_DTYPE)


def test_stale_graph_is_refused_and_failed_capture_is_abandoned(calib):
    """A graph bakes in pointers to the context's tables.  A batch of several frames has geometry tables (frame records,
    per-segment records, tail block table): another batch shape rewrites them, and the graph must be refused afterwards.
    (A run of ONE frame has no table -- its record travels by value -- so single-frame graphs of different sizes coexist.)"""
    import torch
    from lidar_object_detection_amd import synthetic as S
    from lidar_object_detection_amd._native import LpfContext, LpfError, SUMMARY_DTYPE
    _, T, K, W, H = S.default_calibration(calib)
    dev = torch.device("cuda", 0)
    n, M, Bx = 50_000, 4, 5
    sc = S.scene(n, n_masks=M, n_boxes=Bx, seed=91)
    empty = np.zeros((0, 8, 3))
    with LpfContext(0) as ctx:
        ctx.set_camera(T, K, W, H, 0.0, 30.0)
        pts = torch.from_numpy(np.concatenate([sc["points"], sc["points"][:1000]])).to(dev)      # frame 0: the scene, frame 1: a short one
        masks = torch.from_numpy(np.stack([sc["masks"], sc["masks"]])).to(dev)
        o = _outputs(torch, dev, n + 1000, M, Bx, SUMMARY_DTYPE.itemsize)
        o["summary"] = torch.zeros(2 * SUMMARY_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        o["inst_idx"] = torch.empty((2, n), dtype=torch.int64, device=dev)
        torch.cuda.synchronize(dev)
        ctx.set_boxes([sc["corners_velo"], empty])
        step = ctx.make_device_step(pts, np.array([0, n, n + 1000], np.int64), masks_u8=masks, inst_cap=n, **o)
        step()
        ctx.sync()
        ctx.graph_begin()
        step()
        g = ctx.graph_end()
        ctx.graph_launch(g)                                 # fresh: replays
        ctx.sync()
        first = {k: (v[:n] if k in ("uv", "label_bits", "valid_idx") else v) for k, v in o.items()}
        first["inst_idx"] = o["inst_idx"][0]
        _check(first, sc, T, K, W, H, M, Bx, SUMMARY_DTYPE)
        # another batch shape rewrites the tables the graph points at -> the graph must be refused afterwards
        other = ctx.make_device_step(pts, np.array([0, n // 2, n + 1000], np.int64), masks_u8=masks, inst_cap=n, **o)
        other()
        ctx.sync()
        with pytest.raises(LpfError) as ei:
            ctx.graph_launch(g)
        assert ei.value.code == -3 and "stale" in str(ei.value)
        ctx.graph_destroy(g)
        # an error inside a capture (here: a batch shape that needs a table upload, which cannot be captured) abandons the
        # capture instead of leaving the stream in capture mode
        ctx.graph_begin()
        with pytest.raises(LpfError):
            step()                                          # the first shape again: its tables have to travel -> not capturable
        with pytest.raises(LpfError):
            ctx.graph_end()                                 # nothing is being captured any more
        step()                                              # and the context is usable
        ctx.sync()
        _check(first, sc, T, K, W, H, M, Bx, SUMMARY_DTYPE)
        # single frames carry no table: graphs of two different sizes stay valid side by side
        ctx.set_boxes(sc["corners_velo"])
        o1 = _outputs(torch, dev, n, M, Bx, SUMMARY_DTYPE.itemsize)
        o2 = _outputs(torch, dev, n, M, Bx, SUMMARY_DTYPE.itemsize)
        full = ctx.make_device_step(pts[:n], np.array([0, n], np.int64), masks_u8=masks[:1], inst_cap=n, **o1)
        part = ctx.make_device_step(pts[:n // 2], np.array([0, n // 2], np.int64), masks_u8=masks[:1], inst_cap=n, **o2)
        full(); part()
        ctx.sync()
        ctx.graph_begin(); full(); g_full = ctx.graph_end()
        ctx.graph_begin(); part(); g_part = ctx.graph_end()
        for t in o1.values():
            t.zero_()
        torch.cuda.synchronize(dev)
        ctx.graph_launch(g_part)
        ctx.graph_launch(g_full)
        ctx.sync()
        _check(o1, sc, T, K, W, H, M, Bx, SUMMARY_DTYPE)
        ctx.graph_destroy(g_full); ctx.graph_destroy(g_part)


@pytest.mark.parametrize("mode", ["fused", "fused+lent", "fused-pack", "fused-pack+lent"])
def test_software_pipelined_mode_across_state_changes(calib, mode):
    """lpf_set_pipelined(2): the tail of a run rides in the NEXT run's launch and its summaries in the one after.  Whatever
    changes between two runs -- camera window, boxes, masks and their count, batch shape, launch geometry -- the owed work
    must be finished with the state it was queued under.  Nothing is synchronised by the test until every run is queued."""
    import torch
    from lidar_object_detection_amd import synthetic as S
    from lidar_object_detection_amd._native import LpfContext, SUMMARY_DTYPE
    _, T, K, W, H = S.default_calibration(calib)
    dev = torch.device("cuda", 0)
    from conftest import lab_library
    lab = lab_library()                                     # forced geometries only exist in the lab build; without it: by launch size
    plan = [  # (sizes per frame, masks, boxes, depth max, geometry)
        ([40_000, 9_000], 4, 6, 30.0, "auto"), ([40_000, 9_000], 4, 6, 30.0, "auto"), ([40_000, 9_000], 4, 6, 50.0, "auto"),
        ([25_000], 7, 3, 50.0, "auto"), ([25_000], 7, 11, 50.0, "large"), ([70_001, 5, 300], 2, 11, 20.0, "small"),
        ([70_001, 5, 300], 2, 11, 20.0, "large-scan"), ([12_345], 0, 0, 20.0, "auto"), ([12_345], 3, 5, 20.0, "auto")]
    runs = []
    with LpfContext(0, library=lab) as ctx:
        ctx.set_pipelined(mode.split("+")[0])
        for k, (sizes, M, Bx, dmax, geo) in enumerate(plan):
            scenes = [S.scene(max(n, 1), n_masks=max(M, 1), n_boxes=max(Bx, 1), seed=300 + 10 * k + f) for f, n in enumerate(sizes)]
            F, n = len(sizes), int(sum(sizes))
            off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
            pts = torch.from_numpy(np.concatenate([sc["points"][:m] for sc, m in zip(scenes, sizes)])).to(dev)
            o = dict(uv=torch.empty((n, 2), dtype=torch.int32, device=dev), label_bits=torch.empty(n, dtype=torch.int32, device=dev),
                     valid_idx=torch.empty(n, dtype=torch.int64, device=dev), inst_idx=torch.empty((F, max(sizes)), dtype=torch.int64, device=dev),
                     count_mb=torch.zeros(max(F * M * Bx, 1), dtype=torch.int32, device=dev),
                     summary=torch.zeros(F * SUMMARY_DTYPE.itemsize, dtype=torch.uint8, device=dev))
            masks = torch.from_numpy(np.stack([sc["masks"][:M] for sc in scenes])).to(dev) if M else None
            torch.cuda.synchronize(dev)                     # inputs are ready (the context runs on its own stream)
            if lab:
                ctx.set_geometry(geo)
            ctx.set_camera(T, K, W, H, 0.0, dmax)
            if M:
                ctx.set_masks(masks, lend="+lent" in mode)   # lent: read directly by a small launch's tiles, or packed by blocks of a large one
            else:
                ctx.clear_masks()
            if Bx:
                ctx.set_boxes([sc["corners_velo"][:Bx] for sc in scenes])
            else:
                ctx.clear_boxes()
            ctx.run_device(pts, off, inst_cap=max(sizes), **o)
            runs.append((scenes, sizes, M, Bx, dmax, off, o, pts, masks))
        ctx.sync()
    for scenes, sizes, M, Bx, dmax, off, o, pts, masks in runs:
        F = len(sizes)
        sm = np.frombuffer(o["summary"].cpu().numpy().tobytes(), SUMMARY_DTYPE)
        vidx, iidx, cmb = o["valid_idx"].cpu().numpy(), o["inst_idx"].cpu().numpy(), o["count_mb"].cpu().numpy()
        boff = 0
        for f, (sc, nf) in enumerate(zip(scenes, sizes)):
            a = int(off[f])
            limg = orc.pack_masks(sc["masks"][:M], 0, H, W) if M else None
            ref = orc.run(sc["points"][:nf], T, K, W, H, 0.0, dmax, label_img=limg, M=M, corners=sc["corners_velo"][:Bx] if Bx else None,
                          want_float=False)
            assert int(sm[f]["n_valid"]) == ref["n_valid"]
            assert np.array_equal(vidx[a:a + ref["n_valid"]], ref["valid_idx"])
            assert np.array_equal(sm[f]["inst_count"][:M], ref["inst_count"])
            for m in range(M):
                lo, hi = int(sm[f]["inst_off"][m]), int(sm[f]["inst_off"][m + 1])
                assert np.array_equal(iidx[f, lo:hi], ref["inst_lists"][m])
            if M and Bx:
                assert np.array_equal(cmb[M * boff:M * (boff + Bx)].reshape(M, Bx), ref["count_mb"])
                assert np.array_equal(sm[f]["best_box"][:M], ref["best_box"]) and np.array_equal(sm[f]["best_cnt"][:M], ref["best_cnt"])
            boff += Bx


def test_pack_riding_mode_corner_cases(calib):
    """lpf_set_pipelined(4): the pack of lent uint8 masks rides in the run's launch, the run's streaming kernel in the next.
    Corners: the masks' element width changes between consecutive runs (pack and tiles of one launch share it: the pipeline
    is drained first), a run with float masks or an erosion (packed by their own launch), a run without masks, the label image
    read back while a pack is still waiting for its launch, and a host-memory run in between (no launch to ride in)."""
    import torch
    from lidar_object_detection_amd import synthetic as S
    from lidar_object_detection_amd._native import LpfContext, SUMMARY_DTYPE
    _, T, K, W, H = S.default_calibration(calib)
    dev = torch.device("cuda", 0)
    n, Bx = 30_000, 6
    plan = [(5, "u8", 0), (5, "u8", 0), (12, "u8", 0), (20, "u8", 0), (3, "f32", 0), (4, "u8", 1), (0, "u8", 0), (7, "u8", 0), (7, "u8", 0)]
    runs = []
    with LpfContext(0) as ctx:
        ctx.set_pipelined("fused-pack")
        ctx.set_camera(T, K, W, H, 0.0, 30.0)
        sc0 = S.scene(n, n_masks=1, n_boxes=Bx, seed=900)
        ctx.set_boxes([sc0["corners_velo"]])
        for k, (M, kind, erode) in enumerate(plan):
            sc = S.scene(n, n_masks=max(M, 1), n_boxes=Bx, seed=901 + k)
            member = sc["masks"][:M]
            o = _outputs(torch, dev, n, max(M, 1), Bx, SUMMARY_DTYPE.itemsize)
            pts = torch.from_numpy(sc["points"]).to(dev)
            m = None
            if M:
                m = torch.from_numpy(member.astype(np.float32) if kind == "f32" else member).to(dev)
            torch.cuda.synchronize(dev)
            if M:
                ctx.set_masks(m, erode_iters=erode, lend=True)
            else:
                ctx.clear_masks()
            ctx.run_device(pts, np.array([0, n], np.int64), inst_cap=n, **o)
            runs.append((sc, M, erode, o, pts, m))
        ctx.sync()
        for sc, M, erode, o, pts, m in runs:
            sm = np.frombuffer(o["summary"].cpu().numpy().tobytes(), SUMMARY_DTYPE)[0]
            limg = orc.pack_masks(sc["masks"][:M], erode, H, W) if M else None
            ref = orc.run(sc["points"], T, K, W, H, 0.0, 30.0, label_img=limg, M=M, corners=sc0["corners_velo"], want_float=False)
            assert int(sm["n_valid"]) == ref["n_valid"]
            assert np.array_equal(o["valid_idx"][:ref["n_valid"]].cpu().numpy(), ref["valid_idx"])
            assert np.array_equal(o["label_bits"].cpu().numpy().view(np.uint32), ref["label_bits"])
            if M:
                assert np.array_equal(sm["inst_count"][:M], ref["inst_count"])
                assert np.array_equal(o["count_mb"][:M * Bx].cpu().numpy().reshape(M, Bx), ref["count_mb"])
        # a pack still waiting for its launch when the label image is asked for: packed now
        sc = S.scene(n, n_masks=6, n_boxes=Bx, seed=950)
        m = torch.from_numpy(sc["masks"]).to(dev)
        torch.cuda.synchronize(dev)
        ctx.set_masks(m, lend=True)
        assert np.array_equal(ctx.get_label_image()[0], orc.pack_masks(sc["masks"], 0, H, W))
        # ... and a host-memory run right after lent masks were set (it does not go through the step launch)
        ctx.set_masks(m, lend=True)
        try:
            r = ctx.run(sc["points"])
        except Exception as e:                              # the label images rotate with the scratch sets: a loud refusal is fine too
            assert "masks" in str(e)
        else:
            ref = orc.run(sc["points"], T, K, W, H, 0.0, 30.0, label_img=orc.pack_masks(sc["masks"], 0, H, W), M=6,
                          corners=sc0["corners_velo"], want_float=False)
            assert np.array_equal(r["label_bits"], ref["label_bits"]) and np.array_equal(r["count_mb"], ref["count_mb"])


@pytest.mark.parametrize("mode,kind", [("fused", "f32"), ("fused-pack", "u8"), ("fused-pack", "f32")])
def test_sample_frames_as_a_software_pipelined_stream(calib, mode, kind):
    """The 20 KITTI-360 sample frames queued one after the other in a software-pipelined mode -- every frame with its own point
    count, its own masks (lent as the reference has them, float32 0/1, read directly by the tiles of each frame's launch, or as
    uint8) and its own boxes (host corners, as the reference's loop produces them per frame, V3:556-562).  Nothing is synchronised
    until all are queued, and lpf_get_stats shows that the context neither waited nor drained while they were: several real
    frames are in flight together (tiles of one, tail of the one before, summaries of the one before that, in one launch).
    Every frame must equal the reference's own outputs (tests/golden, generated by the reference's functions)."""
    import torch
    from conftest import golden_frames, load_golden, unpack_masks
    from lidar_object_detection_amd._native import LpfContext, SUMMARY_DTYPE
    W, H = int(calib["width"]), int(calib["height"])
    dev = torch.device("cuda", 0)
    tag = "rect5_d50"
    held = []
    for rec in golden_frames()["frames"]:
        g = load_golden(rec["frame"])
        if "u" not in g:
            continue                                        # frame skipped by the reference (no boxes)
        masks = unpack_masks(g, "rect5", H, W)
        M, B, n = masks.shape[0], g["corners_velo"].shape[0], len(g["points"])
        m = torch.from_numpy(masks if kind == "f32" else masks.astype(np.uint8)).to(dev)
        pts = torch.from_numpy(np.ascontiguousarray(g["points"], dtype=np.float32)).to(dev)
        o = dict(uv=torch.empty((n, 2), dtype=torch.int32, device=dev), label_bits=torch.empty(n, dtype=torch.int32, device=dev),
                 valid_idx=torch.empty(n, dtype=torch.int64, device=dev), inst_idx=torch.empty((1, n * max(M, 1)), dtype=torch.int64, device=dev),
                 count_mb=torch.zeros(max(M * B, 1), dtype=torch.int32, device=dev),
                 summary=torch.zeros(SUMMARY_DTYPE.itemsize, dtype=torch.uint8, device=dev))
        held.append((g, M, B, o, pts, m, n))
    assert len(held) >= 15
    torch.cuda.synchronize(dev)
    with LpfContext(0) as ctx:
        ctx.set_pipelined(mode)
        ctx.set_camera(calib["TrVeloToRect"], calib["K"], W, H, 0.0, 50.0)

        def queue_all():
            for g, M, B, o, pts, m, n in held:
                ctx.set_masks(m, lend=True)
                ctx.set_boxes(g["corners_velo"], oriented=True)
                ctx.run_device(pts, np.array([0, n], np.int64), inst_cap=n * max(M, 1), **o)

        for _ in range(4):                                  # warm passes: the four scratch sets and the four box sets grow to the largest
            queue_all()                                     # frame they meet (growing waits); 19 frames: each pass shifts the rotation by 3
        ctx.sync()
        for g, M, B, o, pts, m, n in held:
            for t in o.values():
                t.zero_()
        torch.cuda.synchronize(dev)
        ctx.stats(reset=True)
        queue_all()
        st = ctx.stats()
        ctx.sync()
    assert st["host_waits"] == 0 and st["drains"] == 0 and st["blocking_uploads"] == 0, st
    assert st["step_launches"] == len(held) and st["box_jobs_alone"] == 0, st
    assert st["box_jobs_riding"] == sum(1 for h in held if h[2] > 0), st       # (a frame without boxes has no job)
    for g, M, B, o, pts, m, n in held:
        sm = np.frombuffer(o["summary"].cpu().numpy().tobytes(), SUMMARY_DTYPE)[0]
        uv = o["uv"].cpu().numpy()
        assert np.array_equal(uv[:, 0], np.clip(g["u"], -2**31, 2**31 - 1).astype(np.int32))
        assert np.array_equal(uv[:, 1], np.clip(g["v"], -2**31, 2**31 - 1).astype(np.int32))
        nv = len(g["valid_idx_d50"])
        assert int(sm["n_valid"]) == nv and np.array_equal(o["valid_idx"][:nv].cpu().numpy(), g["valid_idx_d50"])
        assert np.array_equal(sm["inst_count"][:M], g["inst_count_" + tag])
        tot = int(sm["inst_off"][M])
        assert np.array_equal(o["inst_idx"][0, :tot].cpu().numpy(), g["inst_cat_" + tag])
        if M and B:
            assert np.array_equal(o["count_mb"][:M * B].cpu().numpy().reshape(M, B), g["count_mb_" + tag])
