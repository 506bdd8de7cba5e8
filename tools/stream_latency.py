#!/usr/bin/env python3
"""BASELINE.json configs[4]: streaming sequence, 1 M points/frame + 8 eroded 1408x376 masks per
frame (V3 path, depth < 50), the per-frame launch set captured in a hipGraph:
   H2D points (pinned) -> H2D masks (pinned) -> mask pack + 3x3 erosion -> project+label -> scan
   -> lists + box counts -> finalize -> D2H summary + counts
Reports p50 / p95 / p99 wall latency per frame (host launch -> results on the host) against the
100 ms budget of a 10 Hz sensor.  Not a bench line; a parity-checked latency probe.

    python tools/stream_latency.py [--frames 200] [--points 1000000] [--rate-hz 10]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--points", type=int, default=1_000_000)
    ap.add_argument("--distinct", type=int, default=8, help="distinct synthetic frames cycled through")
    ap.add_argument("--no-host-write", action="store_true", help="diagnostic: do not rewrite the pinned input buffers between frames")
    ap.add_argument("--sleep-ms", type=float, default=0.0, help="diagnostic: extra idle time before each frame")
    ap.add_argument("--rate-hz", type=float, default=10.0, help="sensor rate the frames are paced at (0 = back to back)")
    ap.add_argument("--direct", action="store_true", help="issue the launches directly instead of replaying the captured graph")
    args = ap.parse_args()
    import torch
    from lidar_object_detection_amd import synthetic as S
    from lidar_object_detection_amd._native import LpfContext, SUMMARY_DTYPE
    from oracle import cpu_oracle as orc

    _, T, K, W, H = S.default_calibration()
    n, M, B = args.points, 8, 32
    dev = torch.device("cuda", 0)
    scenes = [S.scene(n, M, B, seed=1000 + i) for i in range(args.distinct)]
    h_pts = torch.empty((n, 4), dtype=torch.float32).pin_memory()
    h_masks = torch.empty((1, M, H, W), dtype=torch.uint8).pin_memory()
    h_sum = torch.empty(SUMMARY_DTYPE.itemsize, dtype=torch.uint8).pin_memory()
    h_cnt = torch.empty(M * B, dtype=torch.int32).pin_memory()
    stream = torch.cuda.Stream(dev)
    with torch.cuda.stream(stream):
        d_pts = torch.empty((n, 4), dtype=torch.float32, device=dev)
        d_masks = torch.empty((1, M, H, W), dtype=torch.uint8, device=dev)
        o = dict(uv=torch.empty((n, 2), dtype=torch.int32, device=dev), label_bits=torch.empty(n, dtype=torch.int32, device=dev),
                 valid_idx=torch.empty(n, dtype=torch.int64, device=dev), inst_idx=torch.empty(n, dtype=torch.int64, device=dev),
                 count_mb=torch.zeros(M * B, dtype=torch.int32, device=dev),
                 summary=torch.zeros(SUMMARY_DTYPE.itemsize, dtype=torch.uint8, device=dev))
        ctx = LpfContext(0)
        ctx.set_stream(stream.cuda_stream)
        ctx.set_camera(T, K, W, H, 0.0, 50.0)
        ctx.set_boxes(scenes[0]["corners_velo"])
        step = ctx.make_device_step(d_pts, np.array([0, n], np.int64), masks_u8=d_masks, erode_iters=1, inst_cap=n, **o)

        def frame_work():
            d_pts.copy_(h_pts, non_blocking=True)
            d_masks.copy_(h_masks, non_blocking=True)
            step()
            h_sum.copy_(o["summary"], non_blocking=True)
            h_cnt.copy_(o["count_mb"], non_blocking=True)

        h_pts.copy_(torch.from_numpy(scenes[0]["points"]))
        h_masks.copy_(torch.from_numpy(scenes[0]["masks"])[None])
        frame_work()
        ctx.sync()
        ctx.graph_begin()
        frame_work()
        g = ctx.graph_end()
        lat, t_issue = [], []
        # Back-to-back mode note: the per-frame host copy into pinned memory is a multi-threaded memcpy; in a
        # CPU-quota'd container it can exhaust the cgroup's CFS period and get the process throttled for the rest
        # of it (measured: 5-6 frames of 300 stall ~86 ms with the GPU idle).  Paced at the sensor rate, or with
        # --no-host-write, no frame exceeded 0.5 ms.
        period = 1.0 / args.rate_hz if args.rate_hz > 0 else 0.0
        t_next = time.perf_counter()
        for i in range(args.frames):
            sc = scenes[i % args.distinct]
            if not args.no_host_write:
                h_pts.copy_(torch.from_numpy(sc["points"]))       # "sensor" writes the next scan into pinned memory
                h_masks.copy_(torch.from_numpy(sc["masks"])[None])
            if args.sleep_ms:
                time.sleep(args.sleep_ms * 1e-3)
            if period:
                t_next += period
                d = t_next - time.perf_counter()
                if d > 0:
                    time.sleep(d)
            t0 = time.perf_counter()
            if args.direct:
                frame_work()
            else:
                ctx.graph_launch(g)
            t_issue.append(time.perf_counter() - t0)
            ctx.sync()
            lat.append(time.perf_counter() - t0)
            if i < args.distinct and not args.no_host_write:      # parity of what came back, once per distinct frame
                sm = np.frombuffer(h_sum.numpy().tobytes(), SUMMARY_DTYPE)[0]
                limg = orc.pack_masks(sc["masks"], 1, H, W)
                ref = orc.run(sc["points"], T, K, W, H, 0.0, 50.0, label_img=limg, M=M, corners=scenes[0]["corners_velo"], want_float=False)
                assert int(sm["n_valid"]) == ref["n_valid"] and np.array_equal(sm["inst_count"][:M], ref["inst_count"])
                assert np.array_equal(h_cnt.numpy().reshape(M, B), ref["count_mb"]) and np.array_equal(sm["best_box"][:M], ref["best_box"])
        lat = np.array(lat[5:]) * 1e3
        print(json.dumps({"config": "BASELINE configs[4]: %d frames, %d pts/frame, %d eroded masks, %d boxes, hipGraph per frame incl. H2D/D2H"
                                    % (args.frames, n, M, B),
                          "p50_ms": float(np.percentile(lat, 50)), "p95_ms": float(np.percentile(lat, 95)),
                          "p99_ms": float(np.percentile(lat, 99)), "max_ms": float(lat.max()), "budget_ms": 100.0,
                          "frames_over_1ms": [int(i) + 5 for i in np.nonzero(lat > 1.0)[0][:20]], "n_over_1ms": int((lat > 1.0).sum()),
                          "issue_ms_of_slow_frames": [round(1e3 * t_issue[int(i) + 5], 3) for i in np.nonzero(lat > 1.0)[0][:20]],
                          "mode": "direct launches" if args.direct else "hipGraph replay", "paced_hz": args.rate_hz,
                          "h2d_bytes_per_frame": n * 16 + M * H * W}))
        ctx.graph_destroy(g)
        ctx.close()


if __name__ == "__main__":
    main()
