"""lpf_run_frame: one frame of a stream in ONE C call -- its scan, its lent masks with their rectangles, its annotated boxes (cam-0
corners, filtered and transformed on the device, V3:556-562) and the run.  A software-pipelined stream of the four full-size golden
frames in turn, each call bringing another frame's masks, rectangles and boxes: every frame's results equal the reference-generated
golden vectors, the host never waits, and the results equal those of the four separate calls it replaces."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, check_full

pytestmark = pytest.mark.gpu

NAMES = ("frame_0000000100.npz", "frame_0000001461_full.npz", "frame_0000002098_full.npz", "frame_0000002449_full.npz")
TAG = "rect5_d50"


@pytest.mark.parametrize("mode", [False, "fused", "fused-pack"])
@pytest.mark.parametrize("rects", [True, False])
def test_a_stream_of_frames_one_call_each(calib, mode, rects):
    import torch
    from lidar_object_detection_amd._native import LpfContext, SUMMARY_DTYPE
    W, H = int(calib["width"]), int(calib["height"])
    T, K = np.asarray(calib["TrVeloToRect"]), np.asarray(calib["K"])[:3, :3]
    Tcv = np.linalg.inv(np.asarray(calib["TrVeloToCam"]))
    dev = torch.device("cuda", 0)
    gs = [dict(np.load(os.path.join(GOLDEN, n))) for n in NAMES]
    M = 5
    fr = []
    for g in gs:
        n = len(g["points"])
        masks = np.unpackbits(g["masks_rect5_packed"], axis=-1)[..., :W].astype(np.uint8)
        B = len(g["corners_cam0_raw"])
        fr.append(dict(n=n, B=B, pts=torch.from_numpy(np.ascontiguousarray(g["points"], dtype=np.float32)).to(dev), masks=torch.from_numpy(masks).to(dev),
                       rects=torch.from_numpy(LpfContext.mask_rects(masks)).to(dev), cam0=torch.from_numpy(np.ascontiguousarray(g["corners_cam0_raw"], dtype=np.float64)).to(dev),
                       o=dict(uv=torch.empty((n, 2), dtype=torch.int32, device=dev), label_bits=torch.empty(n, dtype=torch.int32, device=dev),
                              valid_idx=torch.empty(n, dtype=torch.int64, device=dev), inst_idx=torch.empty((1, n), dtype=torch.int64, device=dev),
                              count_mb=torch.zeros(M * B, dtype=torch.int32, device=dev), summary=torch.zeros(SUMMARY_DTYPE.itemsize, dtype=torch.uint8, device=dev))))
    torch.cuda.synchronize(dev)
    with LpfContext(0) as ctx:
        ctx.set_pipelined(mode)
        ctx.set_camera(T, K, W, H, 0.0, 50.0)
        steps = [ctx.make_frame_step(f["pts"], masks_u8=f["masks"], mask_rects=f["rects"] if rects else None, boxes_cam0=f["cam0"], T_cam_to_velo=Tcv,
                                     inst_cap=f["n"], **f["o"]) for f in fr]
        for _ in range(3):                                  # (allocations: every scratch set of the rotation meets every frame size)
            for s in steps:
                s()
        ctx.sync()
        for f in fr:
            for t in f["o"].values():
                t.zero_()
        torch.cuda.synchronize(dev)
        ctx.stats(reset=True)
        for _ in range(3):
            for s in steps:
                s()
        st = ctx.stats()
        ctx.sync()
    if mode:
        assert st["host_waits"] == 0 and st["drains"] == 0 and st["step_launches"] == 12 and st["box_jobs_riding"] == 12, st
    for g, f in zip(gs, fr):
        o, n, B = f["o"], f["n"], f["B"]
        sm = np.frombuffer(o["summary"].cpu().numpy().tobytes(), SUMMARY_DTYPE)[0]
        uv, lab, vidx, inst = o["uv"].cpu().numpy(), o["label_bits"].cpu().numpy().view(np.uint32), o["valid_idx"].cpu().numpy(), o["inst_idx"].cpu().numpy()[0]
        nv = int(sm["n_valid"])
        check_full(g, "u", uv[:, 0], np.int64)
        check_full(g, "v", uv[:, 1], np.int64)
        check_full(g, "valid_idx_d50", vidx[:nv], np.int64)
        check_full(g, "bg_assigned_" + TAG, np.packbits(lab[vidx[:nv]] != 0), np.uint8)
        assert np.array_equal(sm["inst_count"][:M], g["inst_count_" + TAG])
        check_full(g, "inst_cat_" + TAG, inst[:int(sm["inst_off"][M])], np.int64)
        got = o["count_mb"].cpu().numpy().reshape(M, B)
        vis = g["visible_pos"]
        assert np.array_equal(got[:, vis], g["count_mb_" + TAG])
        rest = np.ones(B, bool); rest[vis] = False
        assert not got[:, rest].any()


def test_argument_errors_of_the_one_call_form(calib):
    import ctypes
    import torch
    from lidar_object_detection_amd._native import FrameJob, LpfContext, LpfError
    W, H = int(calib["width"]), int(calib["height"])
    with LpfContext(0) as ctx:
        ctx.set_camera(np.asarray(calib["TrVeloToRect"]), np.asarray(calib["K"])[:3, :3], W, H, 0.0, 50.0)
        j = FrameJob()
        with pytest.raises(LpfError, match="device-mode outputs only"):
            ctx._check(ctx._lib.lpf_run_frame(ctx._h, ctypes.byref(j)))
        j.out.on_device = 1
        j.n_masks = -1
        with pytest.raises(LpfError, match="n_masks"):
            ctx._check(ctx._lib.lpf_run_frame(ctx._h, ctypes.byref(j)))
        assert ctx._lib.lpf_run_frame(ctx._h, None) != 0
        j.n_masks = 0
        j.n_points = 10                                     # points promised, none given: the run's own check
        with pytest.raises(LpfError, match="pts is NULL"):
            ctx._check(ctx._lib.lpf_run_frame(ctx._h, ctypes.byref(j)))
        del torch
