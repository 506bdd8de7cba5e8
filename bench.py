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
counters at the end of the timed region.
"""
import argparse
import json
import os
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


def pmc_traffic(points_per_launch):
    """HBM bytes per launch of the project+label kernel from the committed rocprofv3 --pmc passes
    (FETCH_SIZE / WRITE_SIZE, separate passes, corrected as tools/pmc_summary.py documents), or
    None when no committed pass matches this launch size."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_bench_f8x2M.json")
    if points_per_launch != 8 * 2_000_000 or not os.path.exists(path):
        return None
    with open(path) as f:
        d = json.load(f)
    for k, v in d.items():
        if "lpf_k1_project_t" in k:
            return v["hbm_bytes_per_launch"]
    return None


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
    ap.add_argument("--pipeline", action="store_true",
                    help="overlap a step's short tail kernels (scan, lists, finalize) with the next step's streaming kernel on "
                         "a second stream: more points/s, but the streaming kernel's own duration grows under the contention")
    ap.add_argument("--dist-backend", default="nccl", help="rehearsal only: 'gloo' runs the multi-rank path without RCCL")
    ap.add_argument("--force-device", type=int, default=-1, help="rehearsal only: every rank uses this GPU")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-events", action="store_true", help="skip the second, event-bracketed pass")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world
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
        outs.append(dict(uv=torch.empty((ntot, 2), dtype=torch.int32, device=dev),
                         label_bits=torch.empty(ntot, dtype=torch.int32, device=dev),
                         valid_idx=torch.empty(ntot, dtype=torch.int64, device=dev),
                         inst_idx=torch.empty(ntot, dtype=torch.int64, device=dev),
                         count_mb=torch.zeros(F * N_MASKS * N_BOXES, dtype=torch.int32, device=dev),
                         summary=torch.zeros(F * SUMMARY_DTYPE.itemsize, dtype=torch.uint8, device=dev)))
    frame_off = np.arange(F + 1, dtype=np.int64) * n

    # steps alternate over `streams` contexts (own HIP stream + scratch each), so the short
    # latency-bound kernels of one step overlap the bandwidth-bound kernel of the next
    nstream = max(1, min(args.streams, nbuf))
    nbuf -= nbuf % nstream                                  # buffer b always belongs to context b % nstream
    streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(nstream - 1)]
    ctxs = []
    for st in streams:
        c = LpfContext(local_rank)
        c.set_stream(st.cuda_stream)
        c.set_pipelined(args.pipeline)
        c.set_camera(T, K, W, H, 0.0, DMAX)
        c.set_boxes([sc["corners_velo"] for sc in scenes], oriented=True)   # box parameters resident in HBM
        ctxs.append(c)
    ctx = ctxs[0]

    # one step = K8 mask pack (u8 masks in HBM -> label images) + K1 + scan + K2 + K3, pre-marshalled
    steps_fn = [ctxs[b % nstream].make_device_step(pts_dev[b], frame_off, masks_u8=masks_dev[b], inst_cap=n, **outs[b])
                for b in range(nbuf)]

    def barrier():
        if world > 1:
            dist.barrier()

    def timed(k):
        barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(k):
            steps_fn[i % nbuf]()
        if world > 1:                                               # final aggregate metrics only
            torch.cuda.synchronize(dev)                             # all streams: the last step may be on any of them
            sm = np.frombuffer(outs[(k - 1) % nbuf]["summary"].cpu().numpy().tobytes(), SUMMARY_DTYPE)
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

    torch.cuda.synchronize(dev)                                     # inputs were produced on torch's stream
    for i in range(args.warmup):
        steps_fn[i % nbuf]()
    torch.cuda.synchronize(dev)
    elapsed = timed(args.steps)                                     # pass 1: the reported throughput

    k1_ms, k1_n, elapsed_ev, empty_ms = 0.0, 0, None, 0.0
    if not args.no_events:                                          # pass 2: same steps, HIP events around K1
        empty_ms = float(np.median([c.profile_overhead() for c in ctxs]))   # what an empty bracket measures, live
        for c in ctxs:
            c.profile_enable(True)
            c.profile_read(reset=True)
        elapsed_ev = timed(args.steps)
        for c in ctxs:
            ms_c, n_c = c.profile_read(reset=True)
            k1_ms, k1_n = k1_ms + ms_c, k1_n + n_c
            c.profile_enable(False)

    # the numbers are only reported if the last step's results equal the CPU oracle's (frame 0, rank 0)
    if rank == 0:
        from oracle import cpu_oracle as orc
        b = (args.steps - 1) % nbuf
        sm = np.frombuffer(outs[b]["summary"].cpu().numpy().tobytes(), SUMMARY_DTYPE)
        pts_h = pts_dev[b][:n].cpu().numpy()
        lab = orc.pack_masks(scenes[0]["masks"], 0, H, W)
        o = orc.run(pts_h, T, K, W, H, 0.0, DMAX, label_img=lab, M=N_MASKS, corners=scenes[0]["corners_velo"],
                    want_float=False)
        ok = (int(sm[0]["n_valid"]) == o["n_valid"] and np.array_equal(sm[0]["inst_count"][:N_MASKS], o["inst_count"])
              and np.array_equal(outs[b]["count_mb"][:N_MASKS * N_BOXES].cpu().numpy().reshape(N_MASKS, N_BOXES), o["count_mb"])
              and np.array_equal(outs[b]["valid_idx"][:o["n_valid"]].cpu().numpy(), o["valid_idx"])
              and np.array_equal(outs[b]["uv"][:n].cpu().numpy(), np.stack([o["u"], o["v"]], axis=1))
              and np.array_equal(outs[b]["label_bits"][:n].cpu().numpy().view(np.uint32), o["label_bits"]))
        if not ok:
            raise SystemExit("bench: GPU result differs from the CPU oracle -- refusing to report a number")

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
                       "masks": N_MASKS, "boxes": N_BOXES, "resident_batches_per_gpu": nbuf, "hip_streams": nstream, "tail_kernels_overlap_next_step": bool(args.pipeline),
                       "sharding": "clouds per rank, no data-path collective"},
        }
        if k1_n:
            dur_s = 1e-3 * k1_ms / k1_n                              # mean event bracket around the kernel
            achieved = ALGO_BYTES_PER_POINT * ntot / dur_s / 1e9
            line["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(ntot),
                                "kernel": "lpf_k1_project_t", "avg_us": 1e6 * dur_s, "launches": k1_n,
                                "empty_bracket_us": 1e3 * empty_ms,
                                "algorithmic_bytes_per_launch": ALGO_BYTES_PER_POINT * ntot,
                                "how": "second pass of the same %d steps with hipEvent pairs around the kernel on its stream "
                                       "(ms_per_step of that pass: %.4f). avg_us is the mean bracket as measured: two event "
                                       "records with nothing between them measure empty_bracket_us on the same stream, so the "
                                       "kernel itself lies between avg_us - empty_bracket_us and avg_us (rocprofv3's average of "
                                       "the same command, profiles/, falls inside that interval)"
                                       % (args.steps, 1e3 * elapsed_ev / args.steps)}
        if not args.no_cpu:
            line["cpu_baseline"] = cpu_baseline(scenes[0], T, K, W, H, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    for c in ctxs:
        c.close()


if __name__ == "__main__":
    main()
