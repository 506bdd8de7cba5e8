"""Per-run state in the software-pipelined modes (lpf_set_pipelined 2 / 4).

The reference builds a new box list for EVERY frame (V3:556-562: load_bounding_boxes -> filter_visible_bboxes ->
transform_bboxes_to_velodyne) and every frame has its own point count and masks.  A stream of runs must therefore carry
boxes, masks and batch shape per run without draining the pipeline or waiting for the GPU: the box tables rotate through a
ring of box sets and are built by blocks of the run's own launch, the geometry tables are per scratch set and travel through
a pinned ring, a single frame needs no table at all.  Every run is compared with the CPU oracle; lpf_get_stats proves that
the queued region neither waited nor drained.
"""
import numpy as np
import pytest

from oracle import cpu_oracle as orc
from oracle import numpy_path as npp

pytestmark = pytest.mark.gpu

M = 5


def _plan(shape, k):
    """frame sizes and box counts of run k"""
    if shape == "one_frame_varying":
        return [60_000 + 7_919 * k], [3 + 4 * k]
    if shape == "three_frames_fixed":
        return [30_000, 1_000, 52_000], [6, 9, 4]
    if shape == "two_frames_many_boxes":                    # two and three 64-box words: a ground grid and (a frame or two) four count blocks each
        return [40_000 + 1_500 * k, 25_000], [70 + 3 * k, 0 if k == 5 else 130]
    return [30_000 + 4_097 * k, 1 + k, 70_000 - 5_000 * k], [6 + k, 0 if k == 2 else 9, 4 + 2 * k]     # three_frames_varying


def _outputs(torch, dev, F, n, cap, Btot, summary_bytes):
    return dict(uv=torch.empty((n, 2), dtype=torch.int32, device=dev), label_bits=torch.empty(n, dtype=torch.int32, device=dev),
                valid_idx=torch.empty(n, dtype=torch.int64, device=dev), inst_idx=torch.empty((F, cap), dtype=torch.int64, device=dev),
                count_mb=torch.zeros(max(M * Btot, 1), dtype=torch.int32, device=dev),
                summary=torch.zeros(F * summary_bytes, dtype=torch.uint8, device=dev))


@pytest.mark.parametrize("boxes", ["host_velo", "device_velo_lent", "device_cam0_lent", "device_cam0_copied"])
@pytest.mark.parametrize("shape", ["one_frame_varying", "three_frames_fixed", "three_frames_varying", "two_frames_many_boxes"])
@pytest.mark.parametrize("mode", ["fused", "fused-pack"])
def test_boxes_masks_and_shape_change_every_run(calib, mode, shape, boxes):
    import torch
    from lidar_object_detection_amd import synthetic as S
    from lidar_object_detection_amd._native import LpfContext, SUMMARY_DTYPE
    TrVeloToCam, T, K, W, H = S.default_calibration(calib)
    Tcv = np.linalg.inv(TrVeloToCam)
    dev = torch.device("cuda", 0)
    nruns = 8
    runs = []
    for k in range(nruns):
        sizes, nbox = _plan(shape, k)
        scenes = [S.scene(max(n, 1), n_masks=M, n_boxes=max(b, 1), seed=5000 + 37 * k + f, calib=calib) for f, (n, b) in enumerate(zip(sizes, nbox))]
        F, n = len(sizes), int(sum(sizes))
        off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        boff = np.concatenate([[0], np.cumsum(nbox)]).astype(np.int32)
        pts = torch.from_numpy(np.concatenate([sc["points"][:m] for sc, m in zip(scenes, sizes)])).to(dev)
        masks = torch.from_numpy(np.stack([sc["masks"] for sc in scenes])).to(dev)
        cam0 = np.concatenate([sc["corners_cam0"][:b] for sc, b in zip(scenes, nbox)]).reshape(-1, 8, 3)
        velo = np.concatenate([sc["corners_velo"][:b] for sc, b in zip(scenes, nbox)]).reshape(-1, 8, 3)
        o = _outputs(torch, dev, F, n, max(sizes) * M, int(boff[-1]), SUMMARY_DTYPE.itemsize)
        d_cam0 = torch.from_numpy(np.ascontiguousarray(cam0)).to(dev)
        d_velo = torch.from_numpy(np.ascontiguousarray(velo)).to(dev)
        vis = torch.zeros(max(int(boff[-1]), 1), dtype=torch.uint8, device=dev)
        runs.append(dict(scenes=scenes, sizes=sizes, nbox=nbox, off=off, boff=boff, pts=pts, masks=masks, cam0=cam0, velo=velo, o=o,
                         d_cam0=d_cam0, d_velo=d_velo, vis=vis))
    torch.cuda.synchronize(dev)

    def queue(ctx, r):
        ctx.set_masks(r["masks"], lend=True)
        if boxes == "host_velo":
            ctx.set_boxes([r["velo"][a:b] for a, b in zip(r["boff"][:-1], r["boff"][1:])])
        elif boxes == "device_velo_lent":
            ctx.set_boxes_device(r["d_velo"], r["boff"], lend=True)
        else:
            ctx.set_boxes_cam0_device(r["d_cam0"], r["boff"], Tcv, filter_visible=True, lend=boxes.endswith("lent"), visible=r["vis"])
        ctx.run_device(r["pts"], r["off"], inst_cap=max(r["sizes"]) * M, **r["o"])

    with LpfContext(0) as ctx:
        ctx.set_pipelined(mode)
        ctx.set_camera(T, K, W, H, 0.0, 30.0)
        for _ in range(3):                                  # warm passes: scratch and table buffers grow to size here (growing waits).
            for r in runs:                                  # (three: with three scratch sets in rotation every run meets every set)
                queue(ctx, r)
        ctx.sync()
        for r in runs:
            for t in r["o"].values():
                t.zero_()
        torch.cuda.synchronize(dev)
        ctx.stats(reset=True)
        for r in runs:                                      # the pass that counts: nothing may wait or drain until all runs are queued
            queue(ctx, r)
        st = ctx.stats()
        ctx.sync()
    assert st["host_waits"] == 0 and st["drains"] == 0 and st["blocking_uploads"] == 0, st
    assert st["step_launches"] == nruns and st["box_jobs_riding"] == nruns and st["box_jobs_alone"] == 0, st
    if shape not in ("three_frames_varying", "two_frames_many_boxes"):       # (their frame sizes / box counts change: tables travel)
        assert st["uploads"] == (nruns if boxes == "host_velo" else 0), st      # one frame / an unchanged shape: no table travels
    for r in runs:
        o = r["o"]
        sm = np.frombuffer(o["summary"].cpu().numpy().tobytes(), SUMMARY_DTYPE)
        uv, lab = o["uv"].cpu().numpy(), o["label_bits"].cpu().numpy().view(np.uint32)
        vidx, iidx, cmb = o["valid_idx"].cpu().numpy(), o["inst_idx"].cpu().numpy(), o["count_mb"].cpu().numpy()
        for f, (sc, nf, nb) in enumerate(zip(r["scenes"], r["sizes"], r["nbox"])):
            a, b = int(r["off"][f]), int(r["off"][f + 1])
            b0, b1 = int(r["boff"][f]), int(r["boff"][f + 1])
            corners = r["velo"][b0:b1]
            keep = np.ones(nb, bool)
            if boxes.startswith("device_cam0"):
                keep, velo_ref = npp.prepare_boxes(r["cam0"][b0:b1], np.asarray(K)[:3, :3], W, H, TrVeloToCam)
                assert np.array_equal(r["vis"][b0:b1].cpu().numpy().astype(bool), keep)
                corners = velo_ref
            ref = orc.run(sc["points"][:nf], T, K, W, H, 0.0, 30.0, label_img=orc.pack_masks(sc["masks"], 0, H, W), M=M,
                          corners=corners[keep] if nb else None, want_float=False)
            assert np.array_equal(uv[a:b, 0], ref["u"]) and np.array_equal(uv[a:b, 1], ref["v"])
            assert np.array_equal(lab[a:b], ref["label_bits"])
            assert int(sm[f]["n_valid"]) == ref["n_valid"]
            assert np.array_equal(vidx[a:a + ref["n_valid"]], ref["valid_idx"])
            assert np.array_equal(sm[f]["inst_count"][:M], ref["inst_count"])
            for m in range(M):
                lo, hi = int(sm[f]["inst_off"][m]), int(sm[f]["inst_off"][m + 1])
                assert np.array_equal(iidx[f, lo:hi], ref["inst_lists"][m])
            if nb:
                got = cmb[M * b0:M * b1].reshape(M, nb)
                assert np.array_equal(got[:, keep], ref["count_mb"]) and not got[:, ~keep].any()
                pos = np.flatnonzero(keep)                  # best_box indexes the GIVEN list; the oracle saw the kept boxes only
                want = np.where(ref["best_box"] >= 0, pos[np.maximum(ref["best_box"], 0)] if len(pos) else -1, -1)
                assert np.array_equal(sm[f]["best_box"][:M], want) and np.array_equal(sm[f]["best_cnt"][:M], ref["best_cnt"])


def test_mask_tensor_rewritten_between_steps(calib):
    """make_device_step(lend=False) -- the default -- packs the masks at the call, so a streaming caller may refill the SAME
    mask tensor, in stream order, right after each step call, in every mode.  (With lend=True the tensor of step i is still
    read by the launch of step i+1 or i+2 in the pipelined modes: the caller then has to keep it unchanged.)"""
    import torch
    from lidar_object_detection_amd import synthetic as S
    from lidar_object_detection_amd._native import LpfContext, SUMMARY_DTYPE
    _, T, K, W, H = S.default_calibration(calib)
    dev = torch.device("cuda", 0)
    n, Bx, nsteps = 40_000, 6, 5
    scs = [S.scene(n, n_masks=M, n_boxes=Bx, seed=7100 + k, calib=calib) for k in range(nsteps)]
    stream = torch.cuda.Stream(dev)
    for mode in (False, "fused", "fused-pack"):
        with torch.cuda.stream(stream), LpfContext(0) as ctx:
            ctx.set_stream(stream.cuda_stream)
            ctx.set_pipelined(mode)
            ctx.set_camera(T, K, W, H, 0.0, 30.0)
            ctx.set_boxes(scs[0]["corners_velo"])
            d_masks = torch.zeros((1, M, H, W), dtype=torch.uint8, device=dev)
            pts = [torch.from_numpy(sc["points"]).to(dev) for sc in scs]
            outs = [_outputs(torch, dev, 1, n, n, Bx, SUMMARY_DTYPE.itemsize) for _ in scs]
            steps = [ctx.make_device_step(p, np.array([0, n], np.int64), masks_u8=d_masks, inst_cap=n, **o) for p, o in zip(pts, outs)]
            src = [torch.from_numpy(sc["masks"][None]).to(dev) for sc in scs]
            stream.synchronize()
            for k in range(nsteps):
                d_masks.copy_(src[k], non_blocking=True)    # the same tensor, refilled in stream order before every step
                steps[k]()
            ctx.sync()
            stream.synchronize()
            for sc, o in zip(scs, outs):
                ref = orc.run(sc["points"], T, K, W, H, 0.0, 30.0, label_img=orc.pack_masks(sc["masks"], 0, H, W), M=M,
                              corners=scs[0]["corners_velo"], want_float=False)
                assert np.array_equal(o["label_bits"].cpu().numpy().view(np.uint32), ref["label_bits"]), mode
                assert np.array_equal(o["count_mb"].cpu().numpy().reshape(M, Bx), ref["count_mb"]), mode


def test_mode_switch_with_lent_masks_waiting(calib):
    """lpf_set_pipelined while lent masks are still waiting for their run: they are packed at the switch (a run in the new mode
    must not read a label image nobody wrote), or the run refuses loudly."""
    import torch
    from lidar_object_detection_amd import synthetic as S
    from lidar_object_detection_amd._native import LpfContext, LpfError
    _, T, K, W, H = S.default_calibration(calib)
    dev = torch.device("cuda", 0)
    sc = S.scene(30_000, n_masks=M, n_boxes=4, seed=7300, calib=calib)
    m = torch.from_numpy(sc["masks"]).to(dev)
    torch.cuda.synchronize(dev)
    ref = orc.run(sc["points"], T, K, W, H, 0.0, 30.0, label_img=orc.pack_masks(sc["masks"], 0, H, W), M=M, corners=sc["corners_velo"],
                  want_float=False)
    for first in ("fused", "fused-pack"):
        with LpfContext(0) as ctx:
            ctx.set_camera(T, K, W, H, 0.0, 30.0)
            ctx.set_boxes(sc["corners_velo"])
            ctx.set_pipelined(first)
            ctx.set_masks(m, lend=True)
            ctx.set_pipelined(False)
            try:
                r = ctx.run(sc["points"])
            except LpfError as e:
                assert "masks" in str(e)
            else:
                assert np.array_equal(r["label_bits"], ref["label_bits"]) and np.array_equal(r["count_mb"], ref["count_mb"])


def test_lab_role_clock_sees_every_role_of_a_pipelined_stream(calib):
    """Lab build: lpf_lab_role_clock (tools/role_clock.py) counts the blocks of every role of the step launches -- one summary block,
    one box-job block per 64 boxes, the frame's list and box-count blocks (in four parts: a single frame) and its tiles per run --
    and the results are those of the product library."""
    import torch
    from conftest import lab_library
    from lidar_object_detection_amd import synthetic as S
    from lidar_object_detection_amd._native import LpfContext, SUMMARY_DTYPE
    lab = lab_library()
    if lab is None:
        pytest.skip("liblpf_lab.so has not been built (python -m lidar_object_detection_amd._build lab)")
    TrVeloToCam, T, K, W, H = S.default_calibration(calib)
    Tcv = np.linalg.inv(TrVeloToCam)
    n, M, B = 50_000, 4, 70                                 # 49 segments of 1024 points: 13 list blocks, 2 words of boxes
    sc = S.scene(n, n_masks=M, n_boxes=B, seed=3)
    dev = torch.device("cuda", 0)
    d_pts, d_masks = torch.from_numpy(sc["points"]).to(dev), torch.from_numpy(sc["masks"][None]).to(dev)
    d_cam0 = torch.from_numpy(np.ascontiguousarray(sc["corners_cam0"])).to(dev)
    runs, outs = 12, {}
    for lib in (None, lab):
        o = dict(uv=torch.empty((n, 2), dtype=torch.int32, device=dev), label_bits=torch.empty(n, dtype=torch.int32, device=dev),
                 valid_idx=torch.empty(n, dtype=torch.int64, device=dev), inst_idx=torch.empty((1, n), dtype=torch.int64, device=dev),
                 count_mb=torch.zeros(M * B, dtype=torch.int32, device=dev), summary=torch.zeros(SUMMARY_DTYPE.itemsize, dtype=torch.uint8, device=dev))
        with LpfContext(0, library=lib) as ctx:
            ctx.set_pipelined("fused-pack")
            ctx.set_camera(T, K, W, H, 0.0, 40.0)
            fn = ctx.make_device_step(d_pts, np.array([0, n], np.int64), masks_u8=d_masks, lend=True, boxes_cam0=d_cam0,
                                      box_off=np.array([0, B], np.int32), T_cam_to_velo=Tcv, inst_cap=n, **o)
            if lib is not None:
                assert ctx.role_clock(reset=True) == {}     # switched on; nothing has run
            for _ in range(runs):
                fn()
            clk = ctx.role_clock() if lib is not None else None      # (synchronises: the pipeline is drained)
            ctx.sync()
        outs[lib] = {k: v.cpu().numpy().copy() for k, v in o.items()}
    nv = int(np.frombuffer(outs[None]["summary"].tobytes(), SUMMARY_DTYPE)[0]["n_valid"])
    assert 0 < nv < n
    for k in ("uv", "label_bits", "count_mb", "summary"):
        assert np.array_equal(outs[None][k], outs[lab][k]), k
    assert np.array_equal(outs[None]["valid_idx"][:nv], outs[lab]["valid_idx"][:nv])        # (beyond n_valid: never written)
    nseg = (n + 1023) // 1024
    nlist = (nseg + 3) // 4
    assert clk["summaries"]["blocks"] == runs and clk["box job"]["blocks"] == 2 * runs, clk
    assert clk["lists"]["blocks"] == nlist * runs and clk["box counts"]["blocks"] == nlist * 2 * 4 * runs, clk
    assert clk["project+label tiles"]["blocks"] == 2 * nseg * runs, clk      # 512-point tiles
    for r in clk.values():
        assert 0.0 < r["mean_us"] <= r["longest_us"] < 1e4, clk
