#!/usr/bin/env python3
"""Benchmark of the LiDAR projection + instance point-filter hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path (mask pack -> project+label -> index lists + box
counts -> per-frame summary) over one batch of synthetic clouds of BASELINE.json
configs[2]'s shape: --frames clouds (default 8) of 2 M points, each with 8 disk masks and
32 boxes, V4 clip (depth < 30), processed by one batched launch set.  Inputs are resident
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


PMC_FILE = "r02_pmc_bench_f8x2M.json"


def kernel_source_sha():
    """sha256 (16 hex digits) of the kernel sources: a committed counter pass only describes the kernels it ran."""
    h = hashlib.sha256()
    for name in ("lpf_kernels.hip.h", "lpf_api.hip"):
        with open(os.path.join(ROOT, "lidar_object_detection_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


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


# How a step's kernels are queued (name -> tail overlaps next step, pack on a side stream, CUs of the side streams, text)
MODES = {
    "serial": (False, False, 0, "every kernel of a step on one stream, in order"),
    "pipeline": (True, False, 0, "tail kernels (scan, lists, finalize) of step i on a second stream, overlapping step i+1"),
    "pipeline-pack": (True, True, 0, "pipeline + the mask pack of step i+1 on a third stream"),
    "partition": (True, True, 32, "pipeline-pack with both side streams confined to 32 CUs (4 per XCD)"),
    "fused": ("fused", False, 0, "software pipelining in one launch per step: the tail of step i-1 and the summaries of step i-2 "
                                 "ride among the streaming tiles of step i"),
    "fused-pack": ("fused-pack", False, 0, "software pipelining in one launch per step: the mask pack of step i, the streaming tiles of "
                                           "step i-1, the tail of step i-2 and the summaries of step i-3"),
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


def secondary_lines(torch, dev, local_rank, T, K, W, H):
    """The other BASELINE.json configs on this GPU, same process, each checked before it is timed."""
    from lidar_object_detection_amd import synthetic as S
    from lidar_object_detection_amd._native import LpfContext, SUMMARY_DTYPE
    from oracle import cpu_oracle as orc
    out = {}
    stream = torch.cuda.Stream(dev)

    def sync():
        stream.synchronize()

    # ---- configs[2] literally: ONE synthetic 2 M-point cloud + 8 masks + 32 boxes per launch set ------------------
    with torch.cuda.stream(stream), LpfContext(local_rank) as ctx:
        ctx.set_stream(stream.cuda_stream)
        ctx.set_camera(T, K, W, H, 0.0, DMAX)
        n = N_POINTS
        scs = [S.scene(n, N_MASKS, N_BOXES, seed=7000 + i) for i in range(6)]     # 6 x 56 MB of traffic: past the 256 MiB cache
        ctx.set_boxes(scs[0]["corners_velo"])
        bufs = []
        for sc in scs:
            o = make_outputs(torch, dev, n, 1, n, N_MASKS, N_BOXES, SUMMARY_DTYPE.itemsize)
            bufs.append((torch.from_numpy(sc["points"]).to(dev), torch.from_numpy(sc["masks"][None]).to(dev), o))
        off = np.array([0, n], np.int64)
        fns = [ctx.make_device_step(p_, off, masks_u8=m_, inst_cap=n, **o) for p_, m_, o in bufs]
        sync()
        fns[0]()
        sync()
        o = bufs[0][2]
        ref = orc.run(scs[0]["points"], T, K, W, H, 0.0, DMAX, label_img=orc.pack_masks(scs[0]["masks"], 0, H, W), M=N_MASKS,
                      corners=scs[0]["corners_velo"], want_float=False)
        sm = np.frombuffer(o["summary"].cpu().numpy().tobytes(), SUMMARY_DTYPE)[0]
        ok = (int(sm["n_valid"]) == ref["n_valid"] and np.array_equal(o["valid_idx"][:ref["n_valid"]].cpu().numpy(), ref["valid_idx"])
              and np.array_equal(o["count_mb"].cpu().numpy().reshape(N_MASKS, N_BOXES), ref["count_mb"])
              and np.array_equal(o["uv"].cpu().numpy(), np.stack([ref["u"], ref["v"]], axis=1))
              and np.array_equal(o["label_bits"].cpu().numpy().view(np.uint32), ref["label_bits"]))
        if not ok:
            raise SystemExit("bench secondary configs[2]: GPU result differs from the CPU oracle")
        dt = time_steps(torch, dev, fns, 300, 30, sync)
        ctx.profile_enable(True)
        ctx.profile_read(reset=True)
        time_steps(torch, dev, fns, 100, 0, sync)
        ms, cnt = ctx.profile_read(reset=True)
        ctx.profile_enable(False)
        k1_us = 1e3 * ms / max(cnt, 1)
        # the same clouds as a software-pipelined stream (one launch per cloud: its mask pack, the previous cloud's
        # project+label tiles, the tail of the one before): throughput of a stream of single clouds, not latency
        ctx.set_pipelined("fused-pack")
        for f_ in fns:
            f_()
        ctx.sync()
        sm = np.frombuffer(o["summary"].cpu().numpy().tobytes(), SUMMARY_DTYPE)[0]
        if not (int(sm["n_valid"]) == ref["n_valid"] and np.array_equal(o["count_mb"].cpu().numpy().reshape(N_MASKS, N_BOXES), ref["count_mb"])
                and np.array_equal(o["valid_idx"][:ref["n_valid"]].cpu().numpy(), ref["valid_idx"])):
            raise SystemExit("bench secondary configs[2], pipelined stream: GPU result differs from the CPU oracle")
        dt_p = time_steps(torch, dev, fns, 300, 30, lambda: (ctx.sync(), sync()))
        ctx.set_pipelined(False)
        out["configs2_one_2M_cloud_per_launch"] = {
            "points_per_s": n / dt, "us_per_step": 1e6 * dt, "k1_bracket_us": k1_us,
            "k1_algorithmic_GBps": ALGO_BYTES_PER_POINT * n / (k1_us * 1e-6) / 1e9,
            "step_algorithmic_frac_of_hbm_peak": ALGO_BYTES_PER_POINT * n / dt / 1e9 / HBM_PEAK_GBS,
            "us_per_step_software_pipelined": 1e6 * dt_p, "points_per_s_software_pipelined": n / dt_p,
            "step_algorithmic_frac_of_hbm_peak_software_pipelined": ALGO_BYTES_PER_POINT * n / dt_p / 1e9 / HBM_PEAK_GBS,
            "checked": "n_valid, valid_idx, u, v, label_bits, count_mb == CPU oracle"}
        del bufs, fns

    # ---- configs[3] shape: the 20 sample frames' worth of REAL scan data in one batch (golden frame 100 x 20) -------
    gpath = os.path.join(ROOT, "tests", "golden", "frame_0000000100.npz")
    if os.path.exists(gpath):
        g = np.load(gpath)
        cal = np.load(os.path.join(ROOT, "tests", "golden", "calib_cam0.npz"))
        Tg, Kg, Wg, Hg = np.asarray(cal["TrVeloToRect"]), np.asarray(cal["K"])[:3, :3], int(cal["width"]), int(cal["height"])
        pts = np.ascontiguousarray(g["points"])
        masks = np.unpackbits(g["masks_rect5_packed"], axis=-1)[..., :Wg].astype(np.uint8)
        corners = g["corners_velo"]
        n, M, B, F = len(pts), len(masks), len(corners), 20
        with torch.cuda.stream(stream), LpfContext(local_rank) as ctx:
            ctx.set_stream(stream.cuda_stream)
            ctx.set_camera(Tg, Kg, Wg, Hg, 0.0, 50.0)
            ctx.set_boxes([corners] * F)
            d_pts = torch.from_numpy(np.tile(pts, (F, 1))).to(dev)
            d_masks = torch.from_numpy(np.tile(masks[None], (F, 1, 1, 1))).to(dev)
            o = make_outputs(torch, dev, F * n, F, F * n, M, F * B, SUMMARY_DTYPE.itemsize)
            fn = ctx.make_device_step(d_pts, np.arange(F + 1, dtype=np.int64) * n, masks_u8=d_masks, inst_cap=n, **o)
            sync()
            fn()
            sync()
            cm = o["count_mb"].cpu().numpy().reshape(F, M, B)
            sm = np.frombuffer(o["summary"].cpu().numpy().tobytes(), SUMMARY_DTYPE)
            vi = o["valid_idx"].cpu().numpy()
            ok = all(np.array_equal(cm[f], g["count_mb_rect5_d50"]) and int(sm[f]["n_valid"]) == len(g["valid_idx_d50"])
                     and np.array_equal(vi[f * n:f * n + len(g["valid_idx_d50"])], g["valid_idx_d50"]) for f in range(F))
            if not ok:
                raise SystemExit("bench secondary configs[3]: GPU result differs from the golden vectors of frame 100")
            dt = time_steps(torch, dev, [fn], 500, 30, sync)

            def fused_loop(step_fn, check):
                """the same step in a loop under the software-pipelined mode (a stream of batches: throughput, not latency)"""
                ctx.set_pipelined("fused-pack")
                for _ in range(3):
                    step_fn()
                ctx.sync()
                check()
                t = time_steps(torch, dev, [step_fn], 500, 30, lambda: (ctx.sync(), sync()))
                ctx.set_pipelined(False)
                return t

            def check3():
                cm_ = o["count_mb"].cpu().numpy().reshape(F, M, B)
                if not all(np.array_equal(cm_[f], g["count_mb_rect5_d50"]) for f in range(F)):
                    raise SystemExit("bench secondary configs[3], pipelined: GPU result differs from the golden vectors of frame 100")
            dtf = fused_loop(fn, check3)
            out["configs3_20_real_frames_one_batch"] = {
                "points_per_s": F * n / dt, "us_per_step": 1e6 * dt, "frames": F, "points_per_frame": n, "masks": M, "boxes": B,
                "us_per_step_software_pipelined": 1e6 * dtf, "points_per_s_software_pipelined": F * n / dtf,
                "checked": "count_mb, n_valid, valid_idx of every frame == tests/golden/frame_0000000100.npz (reference functions)"}
            # ... and the same frame alone (configs[1]): launch-bound
            o1 = make_outputs(torch, dev, n, 1, n, M, B, SUMMARY_DTYPE.itemsize)
            ctx.set_boxes(corners)
            fn1 = ctx.make_device_step(d_pts[:n], np.array([0, n], np.int64), masks_u8=d_masks[:1], inst_cap=n, **o1)
            fn1()
            sync()
            if not np.array_equal(o1["count_mb"].cpu().numpy().reshape(M, B), g["count_mb_rect5_d50"]):
                raise SystemExit("bench secondary configs[1]: GPU result differs from the golden vectors of frame 100")
            dt1 = time_steps(torch, dev, [fn1], 1000, 30, sync)

            def check1():
                if not np.array_equal(o1["count_mb"].cpu().numpy().reshape(M, B), g["count_mb_rect5_d50"]):
                    raise SystemExit("bench secondary configs[1], pipelined: GPU result differs from the golden vectors of frame 100")
            dt1f = fused_loop(fn1, check1)
            out["configs1_frame100_device_resident"] = {"points_per_s": n / dt1, "us_per_frame": 1e6 * dt1, "points": n,
                                                        "us_per_frame_in_a_software_pipelined_stream": 1e6 * dt1f,
                                                        "checked": "count_mb == golden"}

    # ---- configs[4]: stream of 1 M-point frames + 8 masks eroded once, H2D + kernels + D2H in one hipGraph per frame ----
    n, M, B = 1_000_000, 8, 32
    scs = [S.scene(n, M, B, seed=1000 + i) for i in range(4)]
    h_pts = torch.empty((n, 4), dtype=torch.float32).pin_memory()
    h_masks = torch.empty((1, M, H, W), dtype=torch.uint8).pin_memory()
    h_sum = torch.empty(SUMMARY_DTYPE.itemsize, dtype=torch.uint8).pin_memory()
    h_cnt = torch.empty(M * B, dtype=torch.int32).pin_memory()
    with torch.cuda.stream(stream), LpfContext(local_rank) as ctx:
        ctx.set_stream(stream.cuda_stream)
        ctx.set_camera(T, K, W, H, 0.0, 50.0)
        ctx.set_boxes(scs[0]["corners_velo"])
        d_pts = torch.empty((n, 4), dtype=torch.float32, device=dev)
        d_masks = torch.empty((1, M, H, W), dtype=torch.uint8, device=dev)
        o = make_outputs(torch, dev, n, 1, n, M, B, SUMMARY_DTYPE.itemsize)
        step = ctx.make_device_step(d_pts, np.array([0, n], np.int64), masks_u8=d_masks, erode_iters=1, inst_cap=n, **o)

        def frame_work():
            d_pts.copy_(h_pts, non_blocking=True)
            d_masks.copy_(h_masks, non_blocking=True)
            step()
            h_sum.copy_(o["summary"], non_blocking=True)
            h_cnt.copy_(o["count_mb"], non_blocking=True)

        h_pts.copy_(torch.from_numpy(scs[0]["points"]))
        h_masks.copy_(torch.from_numpy(scs[0]["masks"])[None])
        frame_work()
        ctx.sync()
        ctx.graph_begin()
        frame_work()
        gr = ctx.graph_end()
        lat = []
        for i in range(104):
            sc = scs[i % len(scs)]
            h_pts.copy_(torch.from_numpy(sc["points"]))
            h_masks.copy_(torch.from_numpy(sc["masks"])[None])
            time.sleep(0.002)                                # a sensor does not deliver frames back to back
            t0 = time.perf_counter()
            ctx.graph_launch(gr)
            ctx.sync()
            lat.append(time.perf_counter() - t0)
            if i < len(scs):
                sm = np.frombuffer(h_sum.numpy().tobytes(), SUMMARY_DTYPE)[0]
                ref = orc.run(sc["points"], T, K, W, H, 0.0, 50.0, label_img=orc.pack_masks(sc["masks"], 1, H, W), M=M,
                              corners=scs[0]["corners_velo"], want_float=False)
                if not (int(sm["n_valid"]) == ref["n_valid"] and np.array_equal(sm["inst_count"][:M], ref["inst_count"])
                        and np.array_equal(h_cnt.numpy().reshape(M, B), ref["count_mb"]) and np.array_equal(sm["best_box"][:M], ref["best_box"])):
                    raise SystemExit("bench secondary configs[4]: GPU result differs from the CPU oracle")
        ctx.graph_destroy(gr)
        lat = 1e3 * np.array(lat[4:])
        out["configs4_stream_hipgraph_per_frame"] = {
            "p50_ms": float(np.percentile(lat, 50)), "p95_ms": float(np.percentile(lat, 95)), "max_ms": float(lat.max()),
            "frames": int(len(lat)), "points_per_frame": n, "masks_eroded_once": M, "boxes": B, "budget_ms_at_10Hz": 100.0,
            "includes": "H2D of points + masks (pinned), mask pack + erosion, project+label, lists + box counts, finalize, D2H of counts + summary",
            "checked": "n_valid, inst_count, count_mb, best_box of the first 4 frames == CPU oracle"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--points", type=int, default=N_POINTS, help="points per cloud")
    ap.add_argument("--frames", type=int, default=8, help="clouds per step (one batched launch)")
    ap.add_argument("--buffers", type=int, default=4, help="distinct resident batches cycled through")
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams (one context each) the steps alternate over; 2 overlaps the short kernels of one "
                         "step with the streaming kernel of the next (higher points/s, longer per-kernel durations)")
    ap.add_argument("--mode", default=DEFAULT_MODE, choices=sorted(MODES),
                    help="how a step's kernels are queued: " + "; ".join("%s = %s" % (k, v[3]) for k, v in sorted(MODES.items())))
    ap.add_argument("--pipeline", action="store_true", help="same as --mode pipeline")
    ap.add_argument("--side-cus", type=int, default=-1, help="override the mode's CU partition (multiple of 8, 0 = none)")
    ap.add_argument("--exclusive", action="store_true", help="with a CU partition: the main stream runs on the other CUs only")
    ap.add_argument("--dist-backend", default="nccl", help="rehearsal only: 'gloo' runs the multi-rank path without RCCL")
    ap.add_argument("--force-device", type=int, default=-1, help="rehearsal only: every rank uses this GPU")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-events", action="store_true", help="skip the second, event-bracketed pass")
    ap.add_argument("--no-secondary", action="store_true", help="skip the other BASELINE configs (N = 1 only anyway)")
    ap.add_argument("--dry-launch", action="store_true", help="print the launcher command of --gpus N and exit (tests)")
    ap.add_argument("--lab", default="", help="diagnosis only (no oracle check, never a reported number): 'nolists', 'noboxes' or 'nomasks'")
    args = ap.parse_args()
    if args.pipeline:
        args.mode = "pipeline"

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args, sys.argv[1:], dry=args.dry_launch)

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

    _, T, K, W, H = S.default_calibration()
    n, F = args.points, args.frames
    nbuf = max(1, args.buffers)
    ntot = n * F
    # F distinct seeded clouds per rank (rank r, frame f -> seed 1000*r + f); the further resident
    # batches are GPU-side permutations of them (distinct addresses, same statistics)
    scenes = [S.scene(n, N_MASKS, N_BOXES, seed=1000 * rank + f) for f in range(F)]
    base_pts = torch.from_numpy(np.concatenate([sc["points"] for sc in scenes], axis=0)).to(dev)
    masks0 = torch.from_numpy(np.stack([sc["masks"] for sc in scenes])).to(dev)           # [F,8,H,W] u8
    pts_dev, masks_dev, outs = [], [], []
    for b in range(nbuf):
        if b == 0:
            pts_dev.append(base_pts)
        else:
            perm = torch.cat([f * n + torch.randperm(n, device=dev) for f in range(F)])
            pts_dev.append(base_pts[perm].contiguous())
            del perm
        masks_dev.append(masks0.clone() if b else masks0)
        outs.append(make_outputs(torch, dev, ntot, F, ntot, N_MASKS, F * N_BOXES, SUMMARY_DTYPE.itemsize))
    frame_off = np.arange(F + 1, dtype=np.int64) * n

    def barrier():
        if world > 1:
            dist.barrier()

    def measure(mode, steps, warmup, events):
        """`steps` timed steps under queueing mode `mode` (+ a second, event-bracketed pass), checked against the oracle."""
        # Contexts run on streams of their own (a CU partition needs that).  Everything above was queued on torch's
        # current stream, and the caching allocator may have carved the output tensors from memory that kernels still
        # queued there read (the index tensors of the gathers): each context gets an explicit device-side edge behind
        # that stream before its first kernel, instead of relying on a device-wide synchronisation ("The
        # bench_s1 fault").
        pipelined, pack_side, side_cus, _ = MODES[mode]
        if args.side_cus >= 0:
            side_cus = args.side_cus
        nstream = max(1, min(args.streams, nbuf))
        nb = nbuf - nbuf % nstream                          # buffer b always belongs to context b % nstream
        ctxs = []
        for _ in range(nstream):
            c = LpfContext(local_rank)
            if side_cus:
                c.set_cu_partition(side_cus, exclusive=args.exclusive)
            c.set_pipelined(pipelined, pack_side=pack_side)
            c.set_camera(T, K, W, H, 0.0, DMAX)
            if args.lab not in ("noboxes", "nowork"):
                c.set_boxes([sc["corners_velo"] for sc in scenes], oriented=True)   # box tables resident in HBM
            c.wait_for_stream(torch.cuda.current_stream(dev).cuda_stream)
            ctxs.append(c)
        # one step = K8 mask pack (u8 masks in HBM -> label images) + project/label + lists + box counts + summaries, pre-marshalled
        steps_fn = [ctxs[b % nstream].make_device_step(pts_dev[b], frame_off, masks_u8=None if args.lab == "nomasks" else masks_dev[b],
                                                       inst_cap=n, **outs[b])
                    for b in range(nb)]

        def drain():                                        # pipelined modes: launch what the last runs still owe, then wait
            for c in ctxs:
                c.sync()
            torch.cuda.synchronize(dev)

        def timed(k):
            barrier()
            drain()
            t0 = time.perf_counter()
            for i in range(k):
                steps_fn[i % nb]()
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
            return float(t.item())

        for i in range(warmup):
            steps_fn[i % nb]()
        drain()
        res = {"mode": mode, "steps": steps, "nbuf": nb, "nstream": nstream, "pipelined": bool(pipelined), "pack_side": bool(pack_side),
               "side_cus": side_cus, "elapsed": timed(steps), "k1_ms": 0.0, "k1_n": 0, "elapsed_ev": None, "empty_ms": 0.0}
        if events:                                          # pass 2: same steps, HIP events around the dominant kernel
            res["empty_ms"] = float(np.median([c.profile_overhead() for c in ctxs]))   # what an empty bracket measures, live
            for c in ctxs:
                c.profile_enable(True)
                c.profile_read(reset=True)
            res["elapsed_ev"] = timed(steps)
            for c in ctxs:
                ms_c, n_c = c.profile_read(reset=True)
                res["k1_ms"], res["k1_n"] = res["k1_ms"] + ms_c, res["k1_n"] + n_c
                c.profile_enable(False)
        # the numbers are only reported if the last step's results equal the CPU oracle's (frame 0, rank 0)
        if rank == 0 and not args.lab:
            from oracle import cpu_oracle as orc
            b = (steps - 1) % nb
            sm = np.frombuffer(outs[b]["summary"].cpu().numpy().tobytes(), SUMMARY_DTYPE)
            pts_h = pts_dev[b][:n].cpu().numpy()
            lab = orc.pack_masks(scenes[0]["masks"], 0, H, W)
            o = orc.run(pts_h, T, K, W, H, 0.0, DMAX, label_img=lab, M=N_MASKS, corners=scenes[0]["corners_velo"],
                        want_float=False)
            ok = (int(sm[0]["n_valid"]) == o["n_valid"] and np.array_equal(sm[0]["inst_count"][:N_MASKS], o["inst_count"])
                  and np.array_equal(outs[b]["count_mb"][:N_MASKS * N_BOXES].cpu().numpy().reshape(N_MASKS, N_BOXES), o["count_mb"])
                  and np.array_equal(outs[b]["valid_idx"][:o["n_valid"]].cpu().numpy(), o["valid_idx"])
                  and np.array_equal(outs[b]["uv"][:n].cpu().numpy(), np.stack([o["u"], o["v"]], axis=1))
                  and np.array_equal(outs[b]["label_bits"][:n].cpu().numpy().view(np.uint32), o["label_bits"])
                  and np.array_equal(sm[0]["best_box"][:N_MASKS], o["best_box"]))
            if not ok:
                raise SystemExit("bench (%s): GPU result differs from the CPU oracle -- refusing to report a number" % mode)
            res["n_valid_batch"] = int(sm["n_valid"].sum())
            res["n_masked_batch"] = int(sm["n_labelled"].sum())
            res["n_list_entries_batch"] = int(sm["inst_count"].sum())
        for c in ctxs:
            c.close()
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
    nbuf, nstream, pipelined, pack_side, side_cus = (main_run[k] for k in ("nbuf", "nstream", "pipelined", "pack_side", "side_cus"))
    del pts_dev, masks_dev, outs, base_pts, masks0

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
                                   "masks + %d 3D boxes, V4 clip depth<30; one launch set per step produces all outputs "
                                   "(u,v,label,valid_idx,instance lists,count_mb,best box)" % (F, n, N_MASKS, N_BOXES),
                       "clouds_per_step_per_gpu": F, "points_per_cloud": n, "points_per_step_per_gpu": ntot,
                       "masks": N_MASKS, "boxes": N_BOXES, "resident_batches_per_gpu": nbuf, "hip_streams": nstream,
                       "mode": args.mode, "tail_kernels_overlap_next_step": bool(pipelined), "mask_pack_on_side_stream": bool(pack_side),
                       "side_stream_cus": side_cus, "main_stream_excludes_them": bool(args.exclusive and side_cus),
                       "step_algorithmic_frac_of_hbm_peak": ALGO_BYTES_PER_POINT * ntot / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
                       "sharding": "clouds per rank, no data-path collective"},
        }
        if k1_n:
            dur_s = 1e-3 * k1_ms / k1_n                              # mean event bracket around the kernel
            achieved = ALGO_BYTES_PER_POINT * ntot / dur_s / 1e9
            step_kernel = args.mode in ("fused", "fused-pack")
            traffic, traffic_note = pmc_traffic(ntot, "lpf_step_t" if step_kernel else "lpf_k1_project_t")
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
                                "kernel": "lpf_step_t (mask pack of this step + project+label tiles of the previous one + tail blocks of the one before)"
                                if args.mode == "fused-pack" else
                                "lpf_step_t (project+label tiles of this step + tail blocks of the previous one)" if args.mode == "fused"
                                else "lpf_k1_project_t", "avg_us": 1e6 * dur_s, "launches": k1_n,
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
            line["cpu_baseline"] = cpu_baseline(scenes[0], T, K, W, H, args.cpu_seconds)
        if world == 1 and not args.no_secondary:
            line["secondary"] = secondary_lines(torch, dev, local_rank, T, K, W, H)
            if serial_run is not None:
                sd = 1e-3 * serial_run["k1_ms"] / max(serial_run["k1_n"], 1)
                line["secondary"]["serial_queueing_same_workload"] = {
                    "us_per_step": 1e6 * serial_run["elapsed"] / serial_run["steps"],
                    "points_per_s": float(ntot) * serial_run["steps"] / serial_run["elapsed"],
                    "k1_bracket_us": 1e6 * sd, "k1_algorithmic_GBps": ALGO_BYTES_PER_POINT * ntot / sd / 1e9,
                    "k1_frac_of_hbm_peak": ALGO_BYTES_PER_POINT * ntot / sd / 1e9 / HBM_PEAK_GBS,
                    "note": "every kernel of a step on one stream, in order (pack, project+label, lists + box counts, summaries): the "
                            "project+label kernel runs alone on the chip here"}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
