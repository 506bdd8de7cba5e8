#!/usr/bin/env python3
"""Benchmark of the LiDAR projection + instance point-filter hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path (box preparation -> mask pack -> project+label -> index
lists + box counts -> per-frame summary) over one batch of synthetic clouds of BASELINE.json
configs[2]'s shape: --frames clouds (default 8) of 2 M points, each with 8 disk masks and
32 boxes, V4 clip (depth < 30), processed by one batched launch set.  Like the reference's
frame loop (V3:556-562) every step brings its OWN boxes: their cam-0 corners are lent in HBM,
filter_visible_bboxes + transform_bboxes_to_velodyne + the table set-up run on the device
inside the step (--static-boxes sets them once instead).  Inputs are resident
in HBM before the timed region; every step touches > 256 MiB and the resident batches are
cycled, so the traffic is real HBM traffic, not Infinity-Cache hits.  The timed region
runs twice: once plain (-> value) and once with HIP events around the project+label
kernel (-> roofline).  Rank 0 prints ONE JSON line.

Multi-GPU: frames/clouds are independent units, so ranks shard them with no data-path
collective (weak scaling); the only exchange is one RCCL all-reduce of the aggregate
counters at the end of the timed region.  `python bench.py --gpus N` without a launcher
starts its own N ranks (a parent that never touches the GPU spawns torch.distributed.run).

With N = 1 the line also carries `secondary`: the other BASELINE.json configs measured in
the same process, each checked against the CPU oracle / the committed golden vectors --
configs[2] literally (one 2 M-point cloud per launch), configs[3]'s shape (20 real frames in
one batch) and configs[4] (10 Hz-style stream, hipGraph per frame, p50 latency).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_POINTS = 2_000_000
N_MASKS = 8
N_BOXES = 32
DMAX = 30.0
ALGO_BYTES_PER_POINT = 28          # 16 B xyzI read + 8 B (u,v) write + 4 B label write
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E 8 TB/s


PMC_FILE = (sorted(f for f in os.listdir(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")) if f.endswith("_pmc_bench_f8x2M.json")) or
            ["r04_pmc_bench_f8x2M.json"])[-1]           # the newest round's committed counter pass


def kernel_source_sha():
    """The id of the kernel sources (csrc/ + include/lpf.h + compiler flags, _build.source_id): what liblpf.so carries as
    lpf_build_id().  A committed counter pass only describes the kernels it ran; a number is only printed for a library built
    from the sources beside it."""
    from lidar_object_detection_amd import _build
    return _build.source_id()


def pmc_traffic(points_per_launch, kernel="lpf_k1_project_t"):
    """(HBM bytes per launch of the project+label kernel, provenance note).  The bytes come from the committed
    rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE, separate passes, corrected as tools/pmc_summary.py documents) and
    are only reported when those passes ran THIS source (sha of csrc/ recorded beside them) at this launch size;
    otherwise None -- a stale figure is worse than none."""
    path = os.path.join(ROOT, "profiles", PMC_FILE)
    if not os.path.exists(path):
        return None, "no committed counter pass (profiles/%s)" % PMC_FILE
    with open(path) as f:
        d = json.load(f)
    meta = d.get("_meta", {})
    if meta.get("points_per_launch") != points_per_launch:
        return None, "profiles/%s was collected at %s points per launch" % (PMC_FILE, meta.get("points_per_launch"))
    if meta.get("kernel_source_sha16") != kernel_source_sha():
        return None, "profiles/%s was collected on kernel sources %s, these are %s: re-run tools/refresh_profiles.sh" % (
            PMC_FILE, meta.get("kernel_source_sha16"), kernel_source_sha())
    # (the step kernel has several instantiations in a run: the one launched most is the steady-state step -- the others are
    #  the pipeline's first launch, which carries only a mask pack, and the drain's)
    # (the step kernel's tile size tells the runs apart: 2048-point tiles "<8," are the software-pipelined step, 1024-point tiles
    #  "<4," are the in-order launch of tiles + box job, and the drain launches of the pipeline)
    hits = [(v.get("launches", 0), k, v) for k, v in d.items() if k != "_meta" and kernel in k]
    if hits:
        _, k, v = max(hits, key=lambda t: t[0])
        return v["hbm_bytes_per_launch"], "profiles/%s (rocprofv3 --pmc, kernel sources %s, %s)" % (PMC_FILE, kernel_source_sha(), k.split("(")[0])
    return None, "profiles/%s has no entry for %s" % (PMC_FILE, kernel)


def usable_cpus(cgroup_root="/sys/fs/cgroup"):
    """CPUs this process may actually use: affinity mask, capped by the cgroup CPU quota when there is one
    (cgroup v2 ``cpu.max`` = "<quota|max> <period>", v1 ``cpu/cpu.cfs_quota_us`` + ``cpu.cfs_period_us``)."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    for path in (os.path.join(cgroup_root, "cpu.max"), os.path.join(cgroup_root, "cpu", "cpu.cfs_quota_us")):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open(os.path.join(cgroup_root, "cpu", "cpu.cfs_period_us")).read())
            if quota not in ("max", "-1"):
                n = max(1, min(n, int(float(quota) / period + 0.5)))
            break
        except Exception:
            continue
    return n


def cpu_baseline(scene, T, K, W, H, budget_s):
    """Reference NumPy statements (oracle/numpy_path.py) on this host, bounded sample, BLAS threads = usable CPUs."""
    from oracle import numpy_path as npp
    cpus = usable_cpus()
    limiter = None
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=cpus)             # more threads than the quota allows only get the process throttled
    except Exception:
        cpus = 1
    n = len(scene["points"])
    reps, t0 = 0, time.perf_counter()
    while True:
        npp.frame_path(scene["points"], T, K, W, H, DMAX, scene["masks"], scene["corners_velo"])
        reps += 1
        el = time.perf_counter() - t0
        if el >= budget_s or reps >= 50:
            break
    if limiter is not None:
        limiter.restore_original_limits()
    out = {"value": n * reps / el, "unit": "points/s", "cores": int(cpus), "kind": "port",
           "sample": "%d x the same %d-point cloud (%d masks, %d boxes), NumPy %s statements of V3:565-592/211-233/344-379 "
                     "(oracle/numpy_path.py), BLAS threads = usable CPUs = %d (os.cpu_count()=%d)"
                     % (reps, n, N_MASKS, N_BOXES, np.__version__, cpus, os.cpu_count())}
    # the single-thread C restatement beside it
    from oracle import cpu_oracle as orc
    lab = orc.pack_masks(scene["masks"], 0, H, W)
    t0 = time.perf_counter()
    r2 = 0
    while r2 < 5:
        orc.run(scene["points"], T, K, W, H, 0.0, DMAX, label_img=lab, M=N_MASKS, corners=scene["corners_velo"],
                want_float=False)
        r2 += 1
    out["c_oracle_1thread_points_per_s"] = n * r2 / (time.perf_counter() - t0)
    # ... and on every usable CPU at once: clouds are independent units, one C call per thread (ctypes drops the GIL)
    from concurrent.futures import ThreadPoolExecutor

    def work(_):
        for _i in range(4):
            orc.run(scene["points"], T, K, W, H, 0.0, DMAX, label_img=lab, M=N_MASKS, corners=scene["corners_velo"],
                    want_float=False, inst_stride=max(n // 8, 1))
        return 4
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cpus) as pool:
        done = sum(pool.map(work, range(cpus)))
    out["c_oracle_allcpus_points_per_s"] = n * done / (time.perf_counter() - t0)
    out["c_oracle_threads"] = cpus
    return out


# How a step's kernels are queued (name -> lpf_set_pipelined argument, text)
MODES = {
    "serial": (False, "every kernel of a step on one stream, in order"),
    "fused": ("fused", "software pipelining in one launch per step: the tail of step i-1 and the summaries of step i-2 "
                       "ride among the streaming tiles of step i"),
    "fused-pack": ("fused-pack", "software pipelining in one launch per step: the mask pack and box set-up of step i, the streaming "
                                 "tiles of step i-1, the tail of step i-2 and the summaries of step i-3"),
}
DEFAULT_MODE = "fused-pack"


def launch_ranks(args, argv, dry=False):
    """`python bench.py --gpus N` with no launcher around it: this parent makes no GPU call; it starts the N ranks
    as a child `python -m torch.distributed.run ... bench.py <same arguments>` and leaves with the child's exit code."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + [a for a in argv if a != "--dry-launch"]
    if dry:
        print(json.dumps({"would_launch": args.gpus, "cmd": cmd}))
        return 0
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def make_outputs(torch, dev, ntot, F, inst_cap_total, M, Btot, summary_bytes):
    return dict(uv=torch.empty((ntot, 2), dtype=torch.int32, device=dev),
                label_bits=torch.empty(ntot, dtype=torch.int32, device=dev),
                valid_idx=torch.empty(ntot, dtype=torch.int64, device=dev),
                inst_idx=torch.empty(inst_cap_total, dtype=torch.int64, device=dev),
                count_mb=torch.zeros(max(M * Btot, 1), dtype=torch.int32, device=dev),
                summary=torch.zeros(F * summary_bytes, dtype=torch.uint8, device=dev))


def time_steps(torch, dev, fns, steps, warmup, sync):
    for i in range(warmup):
        fns[i % len(fns)]()
    sync()
    t0 = time.perf_counter()
    for i in range(steps):
        fns[i % len(fns)]()
    sync()
    return (time.perf_counter() - t0) / steps


def secondary_lines(torch, dev, local_rank, T, K, W, H, use_rects=True):
    """The other BASELINE.json configs on this GPU, same process, each checked before it is timed.  As in the reference's frame
    loop (V3:556-562) every step brings its own boxes -- cam-0 corners lent in HBM, prepared on the device inside the step."""
    from lidar_object_detection_amd import synthetic as S
    from lidar_object_detection_amd._native import LpfContext, SUMMARY_DTYPE
    from oracle import cpu_oracle as orc
    from oracle import numpy_path as npp
    out = {}
    stream = torch.cuda.Stream(dev)
    TrVeloToCam = S.default_calibration()[0]
    Tcv = np.linalg.inv(TrVeloToCam)

    def sync():
        stream.synchronize()

    # ---- configs[2] literally: ONE synthetic 2 M-point cloud + 8 masks + 32 boxes per launch set ------------------
    with torch.cuda.stream(stream), LpfContext(local_rank) as ctx:
        ctx.set_stream(stream.cuda_stream)
        ctx.set_camera(T, K, W, H, 0.0, DMAX)
        n = N_POINTS
        scs = [S.scene(n, N_MASKS, N_BOXES, seed=7000 + i) for i in range(6)]     # 6 x 56 MB of traffic: past the 256 MiB cache
        bufs = []
        for sc in scs:
            o = make_outputs(torch, dev, n, 1, n, N_MASKS, N_BOXES, SUMMARY_DTYPE.itemsize)
            bufs.append((torch.from_numpy(sc["points"]).to(dev), torch.from_numpy(sc["masks"][None]).to(dev), o,
                         torch.from_numpy(np.ascontiguousarray(sc["corners_cam0"])).to(dev),
                         None))
        off = np.array([0, n], np.int64)
        boff = np.array([0, N_BOXES], np.int32)
        fns = [ctx.make_device_step(p_, off, masks_u8=m_, lend=True, boxes_cam0=c_, box_off=boff, T_cam_to_velo=Tcv, inst_cap=n, mask_rects=r_, **o)
               for p_, m_, o, c_, r_ in bufs]
        sync()

        def check2(what):
            for i in (0, len(scs) - 1):
                keep, velo = npp.prepare_boxes(scs[i]["corners_cam0"], K, W, H, TrVeloToCam)
                check_frames(torch, bufs[i][2], bufs[i][0], [scs[i]], [velo], [keep], n, T, K, W, H, DMAX, what)
        for f_ in fns:
            f_()
        sync()
        check2("secondary configs[2]")
        dt = time_steps(torch, dev, fns, 300, 30, sync)
        ctx.profile_enable(True)
        ctx.profile_read(reset=True)
        time_steps(torch, dev, fns, 100, 0, sync)
        ms, cnt = ctx.profile_read(reset=True)
        ctx.profile_enable(False)
        k1_us = 1e3 * ms / max(cnt, 1)
        # the same clouds as a software-pipelined stream (one launch per cloud: its mask pack and box set-up, the previous cloud's
        # project+label tiles, the tail of the one before): throughput of a stream of single clouds, not latency
        ctx.set_pipelined("fused-pack")
        for f_ in fns:
            f_()
        ctx.sync()
        check2("secondary configs[2], pipelined stream")
        ctx.stats(reset=True)
        dt_p = time_steps(torch, dev, fns, 300, 30, lambda: (ctx.sync(), sync()))
        st = ctx.stats()
        ctx.set_pipelined(False)
        out["configs2_one_2M_cloud_per_launch"] = {
            "points_per_s": n / dt, "us_per_step": 1e6 * dt, "k1_bracket_us": k1_us, "boxes_change_every_step": True,
            "k1_algorithmic_GBps": ALGO_BYTES_PER_POINT * n / (k1_us * 1e-6) / 1e9,
            "step_algorithmic_frac_of_hbm_peak": ALGO_BYTES_PER_POINT * n / dt / 1e9 / HBM_PEAK_GBS,
            "us_per_step_software_pipelined": 1e6 * dt_p, "points_per_s_software_pipelined": n / dt_p,
            "step_algorithmic_frac_of_hbm_peak_software_pipelined": ALGO_BYTES_PER_POINT * n / dt_p / 1e9 / HBM_PEAK_GBS,
            "host_waits_and_drains_in_the_pipelined_stream": [st["host_waits"] - 2, st["drains"] - 2],   # (the two syncs of the timing loop itself)
            "checked": "first and last cloud: u, v, label_bits, valid_idx, instance lists, count_mb, best box == CPU oracle (in order and pipelined)"}
        del bufs, fns

    # ---- real scans: the committed golden frames (reference inputs, reference-generated outputs) -------------------------------
    gdir = os.path.join(ROOT, "tests", "golden")
    gpath = os.path.join(gdir, "frame_0000000100.npz")
    if os.path.exists(gpath):
        cal = np.load(os.path.join(gdir, "calib_cam0.npz"))
        Tg, Kg, Wg, Hg = np.asarray(cal["TrVeloToRect"]), np.asarray(cal["K"])[:3, :3], int(cal["width"]), int(cal["height"])
        Tcv_g = np.linalg.inv(np.asarray(cal["TrVeloToCam"]))

        def real_frame(path):
            g = np.load(path)
            return dict(points=np.ascontiguousarray(g["points"], dtype=np.float32),
                        masks=np.unpackbits(g["masks_rect5_packed"], axis=-1)[..., :Wg].astype(np.uint8),
                        cam0=np.ascontiguousarray(g["corners_cam0_raw"], dtype=np.float64), pos=g["visible_pos"],
                        count=g["count_mb_rect5_d50"], n_valid=int(g["valid_idx_d50_len"]) if "valid_idx_d50_len" in g.files else len(g["valid_idx_d50"]),
                        inst_count=g["inst_count_rect5_d50"])

        def batch_of(frames):
            """device inputs + outputs of one batch of real frames, and the checker against their golden vectors"""
            Fb = len(frames)
            sizes = [len(fr["points"]) for fr in frames]
            offb = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
            nb_ = [len(fr["cam0"]) for fr in frames]
            boffb = np.concatenate([[0], np.cumsum(nb_)]).astype(np.int32)
            Mb = frames[0]["masks"].shape[0]
            d = dict(F=Fb, sizes=sizes, off=offb, boff=boffb, M=Mb, ntot=int(offb[-1]), cap=max(sizes),
                     pts=torch.from_numpy(np.concatenate([fr["points"] for fr in frames])).to(dev),
                     masks=torch.from_numpy(np.stack([fr["masks"] for fr in frames])).to(dev),
                     cam0=torch.from_numpy(np.concatenate([fr["cam0"] for fr in frames])).to(dev),
                     rects=torch.from_numpy(LpfContext.mask_rects(np.stack([fr["masks"] for fr in frames]))).to(dev))
            d["o"] = dict(uv=torch.empty((d["ntot"], 2), dtype=torch.int32, device=dev), label_bits=torch.empty(d["ntot"], dtype=torch.int32, device=dev),
                          valid_idx=torch.empty(d["ntot"], dtype=torch.int64, device=dev),
                          inst_idx=torch.empty((Fb, d["cap"]), dtype=torch.int64, device=dev),
                          count_mb=torch.zeros(Mb * int(boffb[-1]), dtype=torch.int32, device=dev),
                          summary=torch.zeros(Fb * SUMMARY_DTYPE.itemsize, dtype=torch.uint8, device=dev))

            def check(what):
                cm = d["o"]["count_mb"].cpu().numpy()
                sm = np.frombuffer(d["o"]["summary"].cpu().numpy().tobytes(), SUMMARY_DTYPE)
                for f, fr in enumerate(frames):
                    b0, b1 = int(boffb[f]), int(boffb[f + 1])
                    c_ = cm[Mb * b0:Mb * b1].reshape(Mb, b1 - b0)
                    rest = np.ones(b1 - b0, bool)
                    rest[fr["pos"]] = False
                    if not (np.array_equal(c_[:, fr["pos"]], fr["count"]) and not c_[:, rest].any() and int(sm[f]["n_valid"]) == fr["n_valid"]
                            and np.array_equal(sm[f]["inst_count"][:Mb], fr["inst_count"])):
                        raise SystemExit("bench %s: frame %d differs from the reference-generated golden vectors" % (what, f))
            d["check"] = check
            return d

        def stepper(ctx, d, rects=False):
            return ctx.make_device_step(d["pts"], d["off"], masks_u8=d["masks"], lend=True, boxes_cam0=d["cam0"], box_off=d["boff"],
                                        T_cam_to_velo=Tcv_g, filter_visible=True, inst_cap=d["cap"], mask_rects=d["rects"] if rects else None, **d["o"])

        def fused_loop(ctx, fns_, checks, reps):
            """the same steps in a loop under the software-pipelined mode (a stream of batches: throughput, not latency)"""
            ctx.set_pipelined("fused-pack")
            for _ in range(2):
                for f_ in fns_:
                    f_()
            ctx.sync()
            for c_ in checks:
                c_()
            ctx.stats(reset=True)
            t = time_steps(torch, dev, fns_, reps, 30, lambda: (ctx.sync(), sync()))
            st_ = ctx.stats()
            ctx.set_pipelined(False)
            return t, [st_["host_waits"] - 2, st_["drains"] - 2]

        f100 = real_frame(gpath)
        n100 = len(f100["points"])
        with torch.cuda.stream(stream), LpfContext(local_rank) as ctx:
            ctx.set_stream(stream.cuda_stream)
            ctx.set_camera(Tg, Kg, Wg, Hg, 0.0, 50.0)
            # configs[3] shape: 20 real frames in one batch (frame 100 x 20)
            d20 = batch_of([f100] * 20)
            fn20 = stepper(ctx, d20)
            sync()
            fn20()
            sync()
            d20["check"]("secondary configs[3]")
            dt = time_steps(torch, dev, [fn20], 500, 30, sync)
            dtf, wd = fused_loop(ctx, [fn20], [lambda: d20["check"]("secondary configs[3], pipelined")], 500)
            out["configs3_20_real_frames_one_batch"] = {
                "points_per_s": 20 * n100 / dt, "us_per_step": 1e6 * dt, "frames": 20, "points_per_frame": n100, "masks": d20["M"],
                "boxes_given_per_frame": len(f100["cam0"]), "boxes_kept_by_filter_visible": len(f100["pos"]), "boxes_change_every_step": True,
                "us_per_step_software_pipelined": 1e6 * dtf, "points_per_s_software_pipelined": 20 * n100 / dtf,
                "host_waits_and_drains_in_the_pipelined_stream": wd,
                "checked": "count_mb (kept boxes; dropped boxes zero), n_valid, inst_count of every frame == tests/golden/frame_0000000100.npz (reference functions)"}
            # ... and the same frame alone (configs[1]): launch-bound
            d1 = batch_of([f100])
            fn1 = stepper(ctx, d1)
            fn1()
            sync()
            d1["check"]("secondary configs[1]")
            dt1 = time_steps(torch, dev, [fn1], 1000, 30, sync)
            dt1f, wd1 = fused_loop(ctx, [fn1], [lambda: d1["check"]("secondary configs[1], pipelined")], 1000)
            out["configs1_frame100_device_resident"] = {"points_per_s": n100 / dt1, "us_per_frame": 1e6 * dt1, "points": n100,
                                                        "boxes_change_every_frame": True,
                                                        "us_per_frame_in_a_software_pipelined_stream": 1e6 * dt1f,
                                                        "host_waits_and_drains_in_the_pipelined_stream": wd1,
                                                        "checked": "count_mb, n_valid, inst_count == golden"}
            del d20, d1, fn20, fn1
            # ---- real scan order at the headline's size: 146 frames (the four full-size golden frames in turn, 16.9 M points) per
            #      step, their 5 masks each, their own annotated boxes (31 / 21 / 186 / 314 per frame, filtered on the device), depth < 50
            full = [f100] + [real_frame(os.path.join(gdir, "frame_%010d_full.npz" % fr)) for fr in (1461, 2098, 2449)
                             if os.path.exists(os.path.join(gdir, "frame_%010d_full.npz" % fr))]
            if len(full) == 4:
                nfr = 146
                dbs = [batch_of([full[(i + s_) % 4] for i in range(nfr)]) for s_ in range(2)]     # two resident batches (2 x 270 MB of points)
                fns = [stepper(ctx, d, use_rects) for d in dbs]      # with the masks' 2D rectangles (lpf_set_mask_rects), as a detector gives them
                sync()
                for f_ in fns:
                    f_()
                sync()
                for d in dbs:
                    d["check"]("secondary real scans at headline size")
                ntot_r = dbs[0]["ntot"]
                dt_s = time_steps(torch, dev, fns, 60, 6, sync)
                ctx.set_pipelined("fused-pack")
                for _ in range(3):
                    for f_ in fns:
                        f_()
                ctx.sync()
                for d in dbs:
                    d["check"]("secondary real scans at headline size, pipelined")
                ctx.stats(reset=True)
                dt_r = time_steps(torch, dev, fns, 200, 20, lambda: (ctx.sync(), sync()))
                st = ctx.stats()
                ctx.profile_enable(True)
                ctx.profile_read(reset=True)
                time_steps(torch, dev, fns, 100, 0, lambda: (ctx.sync(), sync()))
                ms, cnt = ctx.profile_read(reset=True)
                ctx.profile_enable(False)
                dt_plain = None
                if use_rects:                               # the same stream without the masks' rectangles (every mask byte read)
                    fns_plain = [stepper(ctx, d, False) for d in dbs]
                    for _ in range(3):
                        for f_ in fns_plain:
                            f_()
                    ctx.sync()
                    for d in dbs:
                        d["check"]("secondary real scans at headline size, pipelined, no mask rectangles")
                    dt_plain = time_steps(torch, dev, fns_plain, 100, 10, lambda: (ctx.sync(), sync()))
                    del fns_plain
                ctx.set_pipelined(False)
                sm = np.frombuffer(dbs[0]["o"]["summary"].cpu().numpy().tobytes(), SUMMARY_DTYPE)
                nv, nl = int(sm["n_valid"].sum()), int(sm["inst_count"].sum())
                br = 1e-3 * ms / max(cnt, 1)
                # itemised.  With the masks' rectangles the tiles read the masks themselves inside the rectangles: no pack, no label image --
                # per valid point a 4-byte look-up in the frame's candidate grid, per list entry a mask byte, + valid_idx and the lists.
                # Without them: + the masks read and the label images written by the riding pack (5 masks x 1408 x 376 bytes per frame:
                # 2.6 MB of masks next to 3.2 MB of strict point traffic per frame) and the label look-ups.
                rect_bytes = int(sum(int(((r[:, :, 2] - r[:, :, 0]) * (r[:, :, 3] - r[:, :, 1])).sum()) for r in [dbs[0]["rects"].cpu().numpy().astype(np.int64)]))
                item_plain = ALGO_BYTES_PER_POINT * ntot_r + nfr * (dbs[0]["M"] + 1) * Wg * Hg + 12 * nv + 8 * nl
                item = (ALGO_BYTES_PER_POINT * ntot_r + 12 * nv + 9 * nl) if use_rects else item_plain
                out["real_scans_at_headline_size"] = {
                    "frames_per_step": nfr, "points_per_step": ntot_r, "valid_fraction": nv / ntot_r, "masked_list_entries_per_step": nl,
                    "boxes_given_per_step": int(dbs[0]["boff"][-1]), "boxes_change_every_step": True, "mode": "fused-pack",
                    "us_per_step": 1e6 * dt_r, "points_per_s": ntot_r / dt_r, "us_per_step_in_order": 1e6 * dt_s,
                    "step_kernel_bracket_us": 1e6 * br,
                    "strict_frac_of_hbm_peak": ALGO_BYTES_PER_POINT * ntot_r / dt_r / 1e9 / HBM_PEAK_GBS,
                    "strict_frac_of_hbm_peak_step_kernel": ALGO_BYTES_PER_POINT * ntot_r / br / 1e9 / HBM_PEAK_GBS,
                    "itemised_bytes_per_step": item, "itemised_frac_of_hbm_peak": item / dt_r / 1e9 / HBM_PEAK_GBS,
                    "mask_bytes_per_step": nfr * dbs[0]["M"] * Wg * Hg,
                    "mask_rectangles_given": bool(use_rects),
                    "mask_bytes_inside_the_rectangles_per_step": rect_bytes,
                    "us_per_step_without_mask_rectangles": None if dt_plain is None else 1e6 * dt_plain,
                    "itemised_frac_of_hbm_peak_without_mask_rectangles": None if dt_plain is None else item_plain / dt_plain / 1e9 / HBM_PEAK_GBS,
                    "masks_with_rectangles": "the tiles read the lent masks themselves inside their rectangles (candidate grid + exact test): "
                                             "no pack, no label image",
                    "why_below_the_synthetic_headline": "not bytes any more (PMC: 580 MB per launch for 474 strict, 4.3 TB/s of a 5.9 TB/s copy ceiling; "
                                                        "round 3: 743 MB) but the tail riding in the launch: a fifth of a real scan's points are valid "
                                                        "and 1.8 % lie on a car (the synthetic cloud: 4.7 % and 0.6 %), half of these frames carry 186 / 314 "
                                                        "annotated boxes, and the lists / box counts of the run before take block slots from the tiles "
                                                        "(in order: tiles + box job 102, tail 53, summaries 7 us); without the rectangles a real frame "
                                                        "brings 2.6 MB of masks for 3.2 MB of strict point traffic; per-kernel split and counters: "
                                                        "profiles/r04_real146_kernel_stats_*.csv, profiles/r04_pmc_real146.json, DESIGN.md section 8",
                    "host_waits_and_drains_in_the_pipelined_stream": [st["host_waits"] - 2, st["drains"] - 2],
                    "data": "KITTI-360 sample frames 100, 1461, 2098, 2449 in turn (tests/golden: the reference's inputs), real scan order",
                    "checked": "count_mb (kept boxes; dropped boxes zero), n_valid, inst_count of all 146 frames of both batches == the golden vectors (reference functions)"}
                del dbs, fns

    # ---- configs[4]: stream of 1 M-point frames + 8 masks eroded once + 32 boxes, H2D + kernels + D2H in one hipGraph per frame ----
    n, M, B = 1_000_000, 8, 32
    nframes = 600
    scs = [S.scene(n, M, B, seed=1000 + i) for i in range(4)]
    h_pts = torch.empty((n, 4), dtype=torch.float32).pin_memory()
    h_masks = torch.empty((1, M, H, W), dtype=torch.uint8).pin_memory()
    h_box = torch.empty((B, 8, 3), dtype=torch.float64).pin_memory()
    h_sum = torch.empty(SUMMARY_DTYPE.itemsize, dtype=torch.uint8).pin_memory()
    h_cnt = torch.empty(M * B, dtype=torch.int32).pin_memory()
    with torch.cuda.stream(stream), LpfContext(local_rank) as ctx:
        ctx.set_stream(stream.cuda_stream)
        ctx.set_camera(T, K, W, H, 0.0, 50.0)
        d_pts = torch.empty((n, 4), dtype=torch.float32, device=dev)
        d_masks = torch.empty((1, M, H, W), dtype=torch.uint8, device=dev)
        d_box = torch.empty((B, 8, 3), dtype=torch.float64, device=dev)
        o = make_outputs(torch, dev, n, 1, n, M, B, SUMMARY_DTYPE.itemsize)
        step = ctx.make_device_step(d_pts, np.array([0, n], np.int64), masks_u8=d_masks, erode_iters=1, inst_cap=n, **o)
        boff = np.array([0, B], np.int32)

        def frame_work():
            d_pts.copy_(h_pts, non_blocking=True)
            d_masks.copy_(h_masks, non_blocking=True)
            d_box.copy_(h_box, non_blocking=True)
            ctx.set_boxes_device(d_box, boff)               # this frame's boxes (velodyne-frame corners): table set-up inside the graph
            step()
            h_sum.copy_(o["summary"], non_blocking=True)
            h_cnt.copy_(o["count_mb"], non_blocking=True)

        def load(sc):
            h_pts.copy_(torch.from_numpy(sc["points"]))
            h_masks.copy_(torch.from_numpy(sc["masks"])[None])
            h_box.copy_(torch.from_numpy(np.ascontiguousarray(sc["corners_velo"])))

        load(scs[0])
        frame_work()
        ctx.sync()
        ctx.graph_begin()
        frame_work()
        gr = ctx.graph_end()
        lat = []
        t_next = time.perf_counter()
        host_threads = torch.get_num_threads()
        torch.set_num_threads(1)                             # the "sensor's" host copies on one thread: bursts of a multi-threaded memcpy are
                                                             # what exhausts the container's CPU quota (tens of ms of throttling with the GPU idle)
        for i in range(nframes + 4):
            sc = scs[i % len(scs)]
            load(sc)                                         # (the "sensor": a multi-threaded 16 MB host copy into pinned memory)
            t_next += 0.010                                  # frames are paced (100 Hz here, 10 x the sensor's rate): back to back, the host
            d_ = t_next - time.perf_counter()                # copies alone exhaust a CPU-quota'd container's CFS period and the process
            if d_ > 0:                                       # is throttled for tens of ms with the GPU idle (tools/stream_latency.py)
                time.sleep(d_)
            t0 = time.perf_counter()
            ctx.graph_launch(gr)
            ctx.sync()
            lat.append(time.perf_counter() - t0)
            if i < len(scs):
                sm = np.frombuffer(h_sum.numpy().tobytes(), SUMMARY_DTYPE)[0]
                ref = orc.run(sc["points"], T, K, W, H, 0.0, 50.0, label_img=orc.pack_masks(sc["masks"], 1, H, W), M=M,
                              corners=sc["corners_velo"], want_float=False)
                if not (int(sm["n_valid"]) == ref["n_valid"] and np.array_equal(sm["inst_count"][:M], ref["inst_count"])
                        and np.array_equal(h_cnt.numpy().reshape(M, B), ref["count_mb"]) and np.array_equal(sm["best_box"][:M], ref["best_box"])):
                    raise SystemExit("bench secondary configs[4]: GPU result differs from the CPU oracle")
        ctx.graph_destroy(gr)
        torch.set_num_threads(host_threads)
        # the device side of such a frame alone: the same launch set -- box table set-up, mask pack + erosion, project+label, lists +
        # box counts, summaries -- captured WITHOUT the copies (inputs resident in HBM), replayed back to back; and as plain launches
        def device_work():
            ctx.set_boxes_device(d_box, boff)
            step()
        device_work()
        ctx.sync()
        ctx.graph_begin()
        device_work()
        gd = ctx.graph_end()
        for _ in range(20):
            ctx.graph_launch(gd)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(300):
            ctx.graph_launch(gd)
        ctx.sync()
        dev_graph_us = 1e6 * (time.perf_counter() - t0) / 300
        ctx.graph_destroy(gd)
        t0 = time.perf_counter()
        for _ in range(300):
            device_work()
        ctx.sync()
        dev_plain_us = 1e6 * (time.perf_counter() - t0) / 300
        lat = 1e3 * np.array(lat[4:])
        out["configs4_stream_hipgraph_per_frame"] = {
            "p50_ms": float(np.percentile(lat, 50)), "p95_ms": float(np.percentile(lat, 95)), "p99_ms": float(np.percentile(lat, 99)),
            "max_ms": float(lat.max()), "frames_over_1ms": int((lat > 1.0).sum()),
            "frames": int(len(lat)), "paced_hz": 100, "points_per_frame": n, "masks_eroded_once": M, "boxes": B, "boxes_change_every_frame": True,
            "budget_ms_at_10Hz": 100.0,
            "device_side_us_per_frame_graph_replay": dev_graph_us, "device_side_us_per_frame_plain_launches": dev_plain_us,
            "device_side_note": "the frame's kernels alone, inputs resident in HBM, back to back: what is left of the p50 is PCIe -- 16 MB of points, "
                                "4.2 MB of masks and the box corners in, counts and summary out -- and the host's wait",
            "includes": "H2D of points + masks + box corners (pinned), box table set-up, mask pack + erosion, project+label, lists + box counts, "
                        "finalize, D2H of counts + summary",
            "checked": "n_valid, inst_count, count_mb, best_box of the first 4 frames == CPU oracle"}

    # ---- the reference-shaped entry points (what a user of the scripts calls): run_frames on sample frame 100, process_frames over files -------
    if os.path.exists(gpath):
        try:
            out["entry_points"] = entry_point_lines(torch, dev, local_rank, gdir)
        except AssertionError as e:                       # a wrong row is a failed bench, like every other check of this file
            raise SystemExit("bench entry points: %s" % e)
    return out


def entry_point_lines(torch, dev, local_rank, gdir):
    """cvs_erosion.py's frame step as the package's run_frames (dicts of the reference's keys out) on sample frame 100, and
    process_frames (cvs:298-379: files -> master CSV) over 20 real frames -- wall time per call / per frame on this host, results checked
    against the reference-generated golden vectors."""
    from lidar_object_detection_amd import pipeline
    cal = np.load(os.path.join(gdir, "calib_cam0.npz"))
    g = np.load(os.path.join(gdir, "frame_0000000100.npz"))
    W, H = int(cal["width"]), int(cal["height"])
    T = np.asarray(cal["TrVeloToRect"])
    cam = type("Cam", (), {"K": np.asarray(cal["K"]), "width": W, "height": H})()
    pts = np.ascontiguousarray(g["points"], dtype=np.float32)
    masks = np.unpackbits(g["masks_rect5_packed"], axis=-1)[..., :W].astype(np.float32)          # YOLO's masks: float32 0 / 1 (V3:72)
    boxes3d = [{"corners_velo": c.tolist()} for c in g["corners_velo"]]
    colors = pipeline.default_colors(len(masks))
    want = g["stats_points_inside_bbox_rect5_d50"].tolist()
    d_masks, d_pts = torch.from_numpy(masks).to(dev), torch.from_numpy(pts).to(dev)
    res = {}
    for name, p_, m_ in (("host_points_host_masks", pts, masks), ("host_points_masks_on_gpu", pts, d_masks), ("points_and_masks_on_gpu", d_pts, d_masks)):
        item = pipeline.FrameInputs(100, p_, m_, boxes3d, colors)
        ts = []
        for _ in range(40):
            t0 = time.perf_counter()
            r = pipeline.run_frames([item], T, cam, 50.0, 10, True, device=local_rank)[0]
            ts.append(time.perf_counter() - t0)
        assert [d["points_inside_bbox"] for d in r["car_statistics"]] == want and np.array_equal(r["valid_indices"], g["valid_idx_d50"]), "run_frames(%s)" % name
        res[name] = round(1e3 * float(np.median(ts[8:])), 4)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import process_frames_bench
    pf = process_frames_bench.measure(reference_too=False)
    return {"run_frames_frame100_ms_per_call_p50": res,
            "run_frames_checked": "car_statistics.points_inside_bbox and valid_indices == tests/golden/frame_0000000100.npz (reference functions)",
            "process_frames_20_real_frames_ms_per_frame": {"batched": pf["ms_per_frame"]["batched"], "read_ahead": pf["ms_per_frame"]["read_ahead"]},
            "process_frames_checked": pf["checked"], "process_frames_note": pf["note"]}


def check_frames(torch, out, pts_dev_b, scenes, velo_per_frame, keep_per_frame, n, T, K, W, H, dmax, what):
    """Every frame of one finished step against the CPU oracle: pixels, labels, valid_idx, every instance list, box counts
    (a box that filter_visible_bboxes dropped: a zero column), best boxes, summary.  Raises SystemExit on the first difference."""
    from lidar_object_detection_amd._native import SUMMARY_DTYPE
    from oracle import cpu_oracle as orc
    F = len(scenes)
    M = N_MASKS
    sm = np.frombuffer(out["summary"].cpu().numpy().tobytes(), SUMMARY_DTYPE)
    boff = 0
    tot = dict(n_valid=0, n_masked=0, n_list=0)
    for f in range(F):
        pts_h = pts_dev_b[f * n:(f + 1) * n].cpu().numpy()
        keep, velo = keep_per_frame[f], velo_per_frame[f]
        B = len(keep)
        lab = orc.pack_masks(scenes[f]["masks"], 0, H, W)
        o = orc.run(pts_h, T, K, W, H, 0.0, dmax, label_img=lab, M=M, corners=velo[keep], want_float=False)
        a = f * n
        uv = out["uv"][a:a + n].cpu().numpy()
        cm = out["count_mb"][M * boff:M * (boff + B)].cpu().numpy().reshape(M, B)
        pos = np.flatnonzero(keep)
        want_best = np.where(o["best_box"] >= 0, pos[np.maximum(o["best_box"], 0)] if len(pos) else -1, -1)
        ok = (int(sm[f]["n_valid"]) == o["n_valid"] and np.array_equal(sm[f]["inst_count"][:M], o["inst_count"])
              and np.array_equal(uv[:, 0], o["u"]) and np.array_equal(uv[:, 1], o["v"])
              and np.array_equal(out["label_bits"][a:a + n].cpu().numpy().view(np.uint32), o["label_bits"])
              and np.array_equal(out["valid_idx"][a:a + o["n_valid"]].cpu().numpy(), o["valid_idx"])
              and np.array_equal(cm[:, keep], o["count_mb"]) and not cm[:, ~keep].any()
              and np.array_equal(sm[f]["best_box"][:M], want_best) and np.array_equal(sm[f]["best_cnt"][:M], o["best_cnt"])
              and int(sm[f]["n_labelled"]) == int(np.count_nonzero(o["label_bits"])))
        if ok:
            inst = out["inst_idx"][f * n:(f + 1) * n].cpu().numpy()          # inst_cap = n entries per frame
            for m in range(M):
                lo, hi = int(sm[f]["inst_off"][m]), int(sm[f]["inst_off"][m + 1])
                ok = ok and np.array_equal(inst[lo:hi], o["inst_lists"][m])
        if not ok:
            raise SystemExit("bench (%s): frame %d of the checked step differs from the CPU oracle -- refusing to report a number" % (what, f))
        tot["n_valid"] += o["n_valid"]; tot["n_masked"] += int(np.count_nonzero(o["label_bits"])); tot["n_list"] += int(o["inst_count"].sum())
        boff += B
    return tot


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--points", type=int, default=N_POINTS, help="points per cloud")
    ap.add_argument("--frames", type=int, default=8, help="clouds per step (one batched launch)")
    ap.add_argument("--buffers", type=int, default=4, help="distinct resident batches cycled through")
    ap.add_argument("--mode", default=DEFAULT_MODE, choices=sorted(MODES),
                    help="how a step's kernels are queued: " + "; ".join("%s = %s" % (k, v[1]) for k, v in sorted(MODES.items())))
    ap.add_argument("--preheat-ms", type=float, default=300.0,
                    help="keep the GPU busy with untimed steps for this long before the W warm-up steps: after the seconds of host-side "
                         "set-up the GPU's clocks are down, and they take tens of milliseconds of load to come back (the same kernel "
                         "averages 95 us in the first 30 ms and 92 us afterwards; 0 = off)")
    ap.add_argument("--mask-rects", action="store_true", help="also pass the masks' 2D rectangles (lpf_set_mask_rects); measured: no change on this workload, whose pack hides behind the tiles (the real-scan line of `secondary` reports both)")
    ap.add_argument("--static-boxes", action="store_true", help="set the boxes once instead of with every step (the reference's loop "
                                                                "builds a new box list per frame: V3:556-562)")
    ap.add_argument("--dist-backend", default="nccl", help="rehearsal only: 'gloo' runs the multi-rank path without RCCL")
    ap.add_argument("--force-device", type=int, default=-1, help="rehearsal only: every rank uses this GPU")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-events", action="store_true", help="skip the second, event-bracketed pass")
    ap.add_argument("--no-secondary", action="store_true", help="skip the other BASELINE configs (N = 1 only anyway)")
    ap.add_argument("--dry-launch", action="store_true", help="print the launcher command of --gpus N and exit (tests)")
    ap.add_argument("--lab", default="", help="diagnosis only (no oracle check, never a reported number): 'nolists', 'noboxes' or 'nomasks'")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args, sys.argv[1:], dry=args.dry_launch)
    # Which binary is measured: the package's liblpf.so, built from the sources beside it (the loader rebuilds or refuses a stale
    # one).  LPF_LIBRARY may name another build only in a lab run, whose line is marked as such and is never a result.
    if args.lab:
        os.environ["LPF_LAB"] = "1"
    elif os.environ.get("LPF_LIBRARY"):
        raise SystemExit("bench.py: LPF_LIBRARY is set (%s) but this is not a --lab run: refusing to time another library" % os.environ["LPF_LIBRARY"])

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world                                            # the launcher is authoritative
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False")
    if args.force_device >= 0:
        local_rank = args.force_device
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = dev if args.dist_backend == "nccl" else torch.device("cpu")      # where collective payloads live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)          # RCCL over xGMI
        else:
            dist.init_process_group(backend=args.dist_backend)

    from lidar_object_detection_amd import synthetic as S
    from lidar_object_detection_amd._native import LpfContext, SUMMARY_DTYPE
    from oracle import numpy_path as npp

    TrVeloToCam, T, K, W, H = S.default_calibration()
    Tcv = np.linalg.inv(TrVeloToCam)
    n, F = args.points, args.frames
    nbuf = max(1, args.buffers)
    ntot = n * F
    per_step_boxes = not args.static_boxes and args.lab not in ("noboxes", "nowork")
    # F distinct seeded clouds per rank (rank r, frame f -> seed 1000*r + f); the further resident
    # batches are GPU-side permutations of them (distinct addresses, same statistics).  Boxes: every resident batch has its
    # own (seeded) 3D boxes, given as the annotation gives them -- 8 corners in the cam-0 frame.
    scenes = [S.scene(n, N_MASKS, N_BOXES, seed=1000 * rank + f) for f in range(F)]
    base_pts = torch.from_numpy(np.concatenate([sc["points"] for sc in scenes], axis=0)).to(dev)
    masks0 = torch.from_numpy(np.stack([sc["masks"] for sc in scenes])).to(dev)           # [F,8,H,W] u8
    # the masks' 2D rectangles, as a detector hands them out beside its masks (cvs_erosion.py:86-87, 110): outside them the masks are zero
    use_rects = bool(args.mask_rects)
    rects0 = torch.from_numpy(LpfContext.mask_rects(np.stack([sc["masks"] for sc in scenes]))).to(dev)     # [F,8,4] int32
    rects_dev = []
    pts_dev, masks_dev, outs, cam0_host, cam0_dev, velo_ref, keep_ref = [], [], [], [], [], [], []
    for b in range(nbuf):
        if b == 0:
            pts_dev.append(base_pts)
        else:
            perm = torch.cat([f * n + torch.randperm(n, device=dev) for f in range(F)])
            pts_dev.append(base_pts[perm].contiguous())
            del perm
        masks_dev.append(masks0.clone() if b else masks0)
        rects_dev.append(rects0.clone() if b else rects0)
        outs.append(make_outputs(torch, dev, ntot, F, ntot, N_MASKS, F * N_BOXES, SUMMARY_DTYPE.itemsize))
        bb = b if per_step_boxes else 0
        cam = [scenes[f]["corners_cam0"] if bb == 0 else S.synthetic_boxes(N_BOXES, seed=500_000 * bb + 1000 * rank + f)[0] for f in range(F)]
        cam0_host.append(cam)
        cam0_dev.append(torch.from_numpy(np.ascontiguousarray(np.concatenate(cam).reshape(-1, 8, 3))).to(dev))
        prep = [npp.prepare_boxes(c, K, W, H, TrVeloToCam) for c in cam]                   # the reference's own box preparation (restated)
        keep_ref.append([p_[0] for p_ in prep]); velo_ref.append([p_[1] for p_ in prep])
    frame_off = np.arange(F + 1, dtype=np.int64) * n
    box_off = np.arange(F + 1, dtype=np.int32) * N_BOXES

    library = {}                                                       # {"path", "build_id"} of the liblpf.so the contexts loaded

    def barrier():
        if world > 1:
            dist.barrier()

    def measure(mode, steps, warmup, events):
        """`steps` timed steps under queueing mode `mode` (+ a second, event-bracketed pass), checked against the oracle."""
        # The context runs on a stream of its own.  Everything above was queued on torch's current stream, and the caching
        # allocator may have carved the output tensors from memory that kernels still queued there read (the index tensors
        # of the gathers): the context gets an explicit device-side edge behind that stream before its first kernel, instead
        # of relying on a device-wide synchronisation ("The bench_s1 fault").
        ctx = LpfContext(local_rank)
        library.update(ctx.library)
        if not args.lab and library["build_id"] != kernel_source_sha():
            raise SystemExit("bench.py: %s reports build id %s, the sources are %s" % (library["path"], library["build_id"], kernel_source_sha()))
        ctx.set_pipelined(MODES[mode][0])
        ctx.set_camera(T, K, W, H, 0.0, DMAX)
        if not per_step_boxes and args.lab not in ("noboxes", "nowork"):
            ctx.set_boxes_cam0(cam0_host[0], Tcv, filter_visible=True, want_outputs=False)     # box tables resident in HBM
        ctx.wait_for_stream(torch.cuda.current_stream(dev).cuda_stream)
        # one step = box preparation + table set-up (lent cam-0 corners in HBM) + K8 mask pack (lent u8 masks in HBM -> label images)
        # + project/label + lists + box counts + summaries, pre-marshalled
        nb = nbuf
        steps_fn = [ctx.make_device_step(pts_dev[b], frame_off, masks_u8=None if args.lab == "nomasks" else masks_dev[b], lend=True,
                                         boxes_cam0=cam0_dev[b] if per_step_boxes else None, box_off=box_off, T_cam_to_velo=Tcv,
                                         filter_visible=True, inst_cap=n, mask_rects=rects_dev[b] if use_rects and args.lab != "nomasks" else None,
                                         **outs[b])
                    for b in range(nb)]

        def drain():                                        # pipelined modes: launch what the last runs still owe, then wait
            ctx.sync()
            torch.cuda.synchronize(dev)

        def timed(k):
            barrier()
            drain()
            ctx.stats(reset=True)
            t0 = time.perf_counter()
            for i in range(k):
                steps_fn[i % nb]()
            queued = ctx.stats()                            # what the host did while it queued the k steps
            drain()                                         # inside the timed region: nothing of the k steps is left undone
            if world > 1:                                   # final aggregate metrics only
                sm = np.frombuffer(outs[(k - 1) % nb]["summary"].cpu().numpy().tobytes(), SUMMARY_DTYPE)
                agg = torch.tensor([int(sm["n_valid"].sum()), int(sm["n_labelled"].sum()), int(sm["inst_count"].sum()), F],
                                   dtype=torch.int64, device=cdev)
                dist.all_reduce(agg, op=dist.ReduceOp.SUM)
                torch.cuda.synchronize(dev)
            barrier()
            el = time.perf_counter() - t0
            t = torch.tensor([el], dtype=torch.float64, device=cdev)
            if world > 1:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item()), queued

        if args.preheat_ms > 0:                             # clocks up (untimed, and not part of the W warm-up steps)
            t_end = time.perf_counter() + 1e-3 * args.preheat_ms
            i = 0
            while time.perf_counter() < t_end:
                steps_fn[i % nb]()
                i += 1
                if i % 64 == 0:
                    ctx.sync()                              # (bounds what is queued ahead)
            drain()
        for i in range(warmup):
            steps_fn[i % nb]()
        drain()
        el, queued = timed(steps)
        res = {"mode": mode, "steps": steps, "nbuf": nb, "elapsed": el, "queued": queued, "k1_ms": 0.0, "k1_n": 0, "elapsed_ev": None, "empty_ms": 0.0}
        if events:                                          # pass 2: same steps, HIP events around the dominant kernel
            res["empty_ms"] = float(ctx.profile_overhead())           # what an empty bracket measures, live
            ctx.profile_enable(True)
            ctx.profile_read(reset=True)
            res["elapsed_ev"] = timed(steps)[0]
            res["k1_ms"], res["k1_n"] = ctx.profile_read(reset=True)
            ctx.profile_enable(False)
        # the numbers are only reported if EVERY frame of the last step equals the CPU oracle's results (rank 0), instance
        # lists included -- and, with per-step boxes, also the step before it (another batch, other boxes)
        if rank == 0 and not args.lab:
            for back in ((0, 1) if (per_step_boxes and steps > 1 and nb > 1) else (0,)):
                b = (steps - 1 - back) % nb
                tot = check_frames(torch, outs[b], pts_dev[b], scenes, velo_ref[b], keep_ref[b], n, T, K, W, H, DMAX, mode)
                if back == 0:
                    res["n_valid_batch"], res["n_masked_batch"], res["n_list_entries_batch"] = tot["n_valid"], tot["n_masked"], tot["n_list"]
            res["checked"] = "every frame of the last step%s: u, v, label_bits, valid_idx, every instance list, count_mb, best box, summary == CPU oracle" % (
                " and of the one before it (other boxes)" if per_step_boxes and steps > 1 and nb > 1 else "")
        ctx.close()
        return res

    if args.lab in ("nolists", "nowork"):
        for o in outs:
            o["valid_idx"] = None
            o["inst_idx"] = None
    main_run = measure(args.mode, args.steps, args.warmup, not args.no_events)
    # the plain in-order queueing beside it (N = 1 only): what the pipelining buys, and the streaming kernel on its own
    serial_run = None
    if world == 1 and not args.no_secondary and args.mode != "serial" and not args.lab:
        serial_run = measure("serial", min(args.steps, 100), min(args.warmup, 10), True)
    elapsed, k1_ms, k1_n, elapsed_ev, empty_ms = (main_run[k] for k in ("elapsed", "k1_ms", "k1_n", "elapsed_ev", "empty_ms"))
    nbuf = main_run["nbuf"]
    del pts_dev, masks_dev, outs, base_pts, masks0, cam0_dev

    if rank == 0:
        total_points = float(ntot) * args.steps * world
        line = {
            "metric": "LiDAR points/sec projected+instance-labelled",
            "value": total_points / elapsed,
            "unit": "points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[2] shape, batched: %d synthetic clouds x %d points per step, each with %d disk "
                                   "masks + %d 3D boxes, V4 clip depth<30; %s; one launch set per step produces all outputs "
                                   "(u,v,label,valid_idx,instance lists,count_mb,best box)"
                                   % (F, n, N_MASKS, N_BOXES, "every step brings its own boxes (cam-0 corners: filter_visible_bboxes + "
                                      "transform_bboxes_to_velodyne + table set-up on the device, inside the step)" if per_step_boxes
                                      else "boxes set once"),
                       "clouds_per_step_per_gpu": F, "points_per_cloud": n, "points_per_step_per_gpu": ntot,
                       "masks": N_MASKS, "boxes": N_BOXES, "boxes_change_every_step": bool(per_step_boxes),
                       "masks_change_every_step": True, "mask_rectangles_given": bool(use_rects), "resident_batches_per_gpu": nbuf, "mode": args.mode,
                       "clock_preheat_ms_before_the_warmup_steps": args.preheat_ms,
                       "host_while_queueing_the_timed_steps": main_run["queued"],
                       "step_algorithmic_frac_of_hbm_peak": ALGO_BYTES_PER_POINT * ntot / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
                       "checked": main_run.get("checked"),
                       "sharding": "clouds per rank, no data-path collective",
                       "library": {"path": os.path.relpath(library.get("path", "?"), ROOT), "build_id": library.get("build_id"),
                                   "sources_id": kernel_source_sha()},
                       "ranks": {"world_size": world, "backend": (dist.get_backend() if world > 1 else None),
                                 "dist_world_size": (dist.get_world_size() if world > 1 else 1)}},
        }
        if k1_n:
            dur_s = 1e-3 * k1_ms / k1_n                              # mean event bracket around the kernel
            achieved = ALGO_BYTES_PER_POINT * ntot / dur_s / 1e9
            step_kernel = args.mode in ("fused", "fused-pack")
            traffic, traffic_note = pmc_traffic(ntot, "lpf_step_t<8" if step_kernel else ("lpf_step_t<4" if per_step_boxes else "lpf_k1_project_t"))
            # SURVEY 8(d) asks for both figures: the strict 28 B per input point (-> achieved, frac) and the itemised total
            # of what this launch produces: + 4 B label-image gather per valid point and, when the launch carries the list
            # blocks too (the fused step), + 8 B valid_idx per valid point + 8 B per instance-list entry; when it carries the
            # mask pack as well (fused-pack), + the masks read and the label images written
            itemised = None
            if "n_valid_batch" in main_run:
                itemised = ALGO_BYTES_PER_POINT * ntot + 4 * main_run["n_valid_batch"]
                if step_kernel:
                    itemised += 8 * main_run["n_valid_batch"] + 8 * main_run["n_list_entries_batch"]
                if args.mode == "fused-pack":
                    itemised += F * (N_MASKS + 1) * W * H
            line["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_note,
                                "kernel": "lpf_step_t (mask pack + box set-up of this step + project+label tiles of the previous one + tail blocks of the one before)"
                                if args.mode == "fused-pack" else
                                "lpf_step_t (project+label tiles + box set-up of this step + tail blocks of the previous one)" if args.mode == "fused"
                                else ("lpf_step_t (project+label tiles + this step's box set-up, in order)" if per_step_boxes else "lpf_k1_project_t"),
                                "avg_us": 1e6 * dur_s, "launches": k1_n,
                                "empty_bracket_us": 1e3 * empty_ms,
                                "algorithmic_bytes_per_launch": ALGO_BYTES_PER_POINT * ntot,
                                "algorithmic_bytes_itemised": itemised,
                                "achieved_itemised": itemised / dur_s / 1e9 if itemised else None,
                                "how": "second pass of the same %d steps with hipEvent pairs around the kernel on its stream "
                                       "(ms_per_step of that pass: %.4f). avg_us is the mean bracket as measured: two event "
                                       "records with nothing between them measure empty_bracket_us on the same stream, so the "
                                       "kernel itself lies between avg_us - empty_bracket_us and avg_us (rocprofv3's average of "
                                       "the same command, profiles/, falls inside that interval)"
                                       % (args.steps, 1e3 * elapsed_ev / args.steps)}
        if args.lab:
            line["metric"] = "LAB RUN (%s): not a benchmark result" % args.lab
        if not args.no_cpu:
            line["cpu_baseline"] = cpu_baseline(dict(scenes[0], corners_velo=velo_ref[0][0][keep_ref[0][0]]), T, K, W, H, args.cpu_seconds)
        if world == 1 and not args.no_secondary:
            line["secondary"] = secondary_lines(torch, dev, local_rank, T, K, W, H)
            if serial_run is not None:
                sd = 1e-3 * serial_run["k1_ms"] / max(serial_run["k1_n"], 1)
                line["secondary"]["serial_queueing_same_workload"] = {
                    "us_per_step": 1e6 * serial_run["elapsed"] / serial_run["steps"],
                    "points_per_s": float(ntot) * serial_run["steps"] / serial_run["elapsed"],
                    "k1_bracket_us": 1e6 * sd, "k1_algorithmic_GBps": ALGO_BYTES_PER_POINT * ntot / sd / 1e9,
                    "k1_frac_of_hbm_peak": ALGO_BYTES_PER_POINT * ntot / sd / 1e9 / HBM_PEAK_GBS,
                    "boxes_change_every_step": bool(per_step_boxes),
                    "note": "every kernel of a step on one stream, in order (box job, pack, project+label, lists + box counts, summaries): the "
                            "project+label kernel runs alone on the chip here"}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
