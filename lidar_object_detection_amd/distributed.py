"""Frame-sharded multi-GPU driver: one process per GPU, ``torch.distributed`` (backend
"nccl" = RCCL over xGMI on ROCm; "gloo" for CPU tests).

Frames are independent units of the reference's loop (``for file in available_files``,
V3:541 / cvs_erosion.py:320), so ranks take frames round-robin and run the whole hot path
locally with no data-path collective.  The only exchanges are at the end:
  * one all-reduce (SUM) of an int64 vector of aggregate counters, one MIN and one MAX
    all-reduce of the matched cars' inside percentage in integer hundredths (exact, because
    the CSV rounds to 2 decimals, cvs_erosion.py:250-251) -- < 100 bytes, latency-bound;
  * an all-gather of the per-car rows, as padded int64 vectors (floats by their bit patterns), so rank 0 can write the
    master CSV in frame order.
The reference has no distributed code; this reproduces its single-process outputs.
"""
import numpy as np

AGG_FIELDS = ("frames_with_rows", "n_cars", "n_matched", "sum_total", "sum_inside", "sum_outside",
              "sum_total_matched", "sum_inside_pct_hundredths_matched")
_BIG = np.int64(1) << 40


def shard_frames(frames, rank, world_size):
    """Round-robin over the ascending frame list: position p -> rank p % world_size."""
    return list(frames)[rank::world_size]


def local_aggregates(rows_by_frame):
    """rows_by_frame: {frame: [stat dicts]}.  Returns (sum vector int64[8], min, max) for this rank."""
    v = np.zeros(len(AGG_FIELDS), np.int64)
    lo, hi = _BIG, -_BIG
    for frame, rows in rows_by_frame.items():
        if rows:
            v[0] += 1
        for r in rows:
            matched = r["matched_bbox_id"] >= 0
            v[1] += 1
            v[2] += int(matched)
            v[3] += r["total_points"]
            v[4] += r["points_inside_bbox"]
            v[5] += r["points_outside_bbox"]
            if matched:
                h = int(round(round(r["inside_percentage"], 2) * 100))
                v[6] += r["total_points"]
                v[7] += h
                lo, hi = min(lo, h), max(hi, h)
    return v, np.int64(lo), np.int64(hi)


def _tensor(arr, device):
    import torch
    return torch.as_tensor(np.asarray(arr, dtype=np.int64), device=device)


def allreduce_aggregates(vec, lo, hi, device="cpu", group=None, ctx=None, rccl_comm=None):
    """SUM / MIN / MAX over all ranks; returns NumPy values every rank can read.  By default through
    torch.distributed (backend "nccl" = RCCL on ROCm); a host that is not a torch program passes its own
    communicator and the context instead and the C ABI does the exchange (lpf_allreduce_metrics)."""
    if rccl_comm is not None:
        if ctx is None:
            raise ValueError("allreduce over a raw RCCL communicator needs the LpfContext of this rank")
        out = ctx.allreduce_metrics(np.asarray(vec, np.int64).copy(), rccl_comm, "sum")
        lo_r = ctx.allreduce_metrics(np.array([lo], np.int64), rccl_comm, "min")
        hi_r = ctx.allreduce_metrics(np.array([hi], np.int64), rccl_comm, "max")
        return out, int(lo_r[0]), int(hi_r[0])
    import torch.distributed as dist
    t = _tensor(vec, device)
    tlo, thi = _tensor([lo], device), _tensor([hi], device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(tlo, op=dist.ReduceOp.MIN, group=group)
        dist.all_reduce(thi, op=dist.ReduceOp.MAX, group=group)
    return t.cpu().numpy(), int(tlo.cpu().item()), int(thi.cpu().item())


ROW_WORDS = 11          # frame, car_id, matched_bbox_id, total, inside, outside, inside % bits, outside % bits, colour r/g/b bits


def encode_rows(rows_by_frame):
    """{frame: [stat dicts]} -> one int64 vector: [n_frames, n_rows, (frame, rows of it) x n_frames, row words x n_rows].
    Floats travel as their bit patterns, so the gathered rows are the local rows, bit for bit."""
    frames = sorted(rows_by_frame)
    rows = [(f, r) for f in frames for r in rows_by_frame[f]]
    out = np.zeros(2 + 2 * len(frames) + ROW_WORDS * len(rows), np.int64)
    out[0], out[1] = len(frames), len(rows)
    for i, f in enumerate(frames):
        out[2 + 2 * i], out[3 + 2 * i] = int(f), len(rows_by_frame[f])
    base = 2 + 2 * len(frames)
    for i, (f, r) in enumerate(rows):
        col = tuple(r.get("color", (0.0, 0.0, 0.0)))[:3]
        fl = np.array([r["inside_percentage"], r["outside_percentage"], col[0], col[1], col[2]], np.float64).view(np.int64)
        out[base + ROW_WORDS * i:base + ROW_WORDS * (i + 1)] = (int(f), r["car_id"], r["matched_bbox_id"], r["total_points"],
                                                                r["points_inside_bbox"], r["points_outside_bbox"], *fl)
    return out


def decode_rows(vec):
    """Inverse of encode_rows (a padded vector is fine: the header says how much of it is payload)."""
    vec = np.asarray(vec, np.int64)
    nf, nr = int(vec[0]), int(vec[1])
    out = {int(vec[2 + 2 * i]): [] for i in range(nf)}
    base = 2 + 2 * nf
    for i in range(nr):
        w = vec[base + ROW_WORDS * i:base + ROW_WORDS * (i + 1)]
        fl = w[6:11].copy().view(np.float64)
        out[int(w[0])].append({"car_id": int(w[1]), "matched_bbox_id": int(w[2]), "total_points": int(w[3]),
                               "points_inside_bbox": int(w[4]), "points_outside_bbox": int(w[5]),
                               "inside_percentage": float(fl[0]), "outside_percentage": float(fl[1]),
                               "color": (float(fl[2]), float(fl[3]), float(fl[4]))})
    return out


def gather_rows(rows_by_frame, device="cpu", group=None):
    """Every rank's {frame: rows} merged and ordered by frame (the CSV row order of the reference): one all-gather of
    the vector lengths, one of the int64 vectors padded to the longest (RCCL all-gather needs equal shapes)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return dict(sorted(decode_rows(encode_rows(rows_by_frame)).items()))
    ws = dist.get_world_size(group)
    mine = encode_rows(rows_by_frame)
    n = _tensor([mine.size], device)
    sizes = [torch.zeros_like(n) for _ in range(ws)]
    dist.all_gather(sizes, n, group=group)
    longest = max(int(t.item()) for t in sizes)
    padded = np.zeros(longest, np.int64)
    padded[:mine.size] = mine
    t = _tensor(padded, device)
    parts = [torch.zeros_like(t) for _ in range(ws)]
    dist.all_gather(parts, t, group=group)
    merged = {}
    for part in parts:
        merged.update(decode_rows(part.cpu().numpy()))
    return dict(sorted(merged.items()))


def format_overall_analysis(vec, lo, hi):
    """The text analyze_master_csv prints (cvs_erosion.py:276-293), from the reduced aggregates."""
    d = dict(zip(AGG_FIELDS, (int(x) for x in vec)))
    lines = ["", "=" * 60, f"{'OVERALL ANALYSIS':^60}", "=" * 60,
             f"Total frames processed: {d['frames_with_rows']}",
             f"Total car detections: {d['n_cars']}",
             f"Successfully matched cars: {d['n_matched']}",
             f"Unmatched cars: {d['n_cars'] - d['n_matched']}"]
    if d["n_cars"]:
        lines.append(f"Average matching rate: {d['n_matched'] / d['n_cars'] * 100:.1f}%")
    if d["n_matched"]:
        lines += ["", "Matched Cars Statistics:",
                  f"Average points per car: {d['sum_total_matched'] / d['n_matched']:.1f}",
                  f"Average inside percentage: {d['sum_inside_pct_hundredths_matched'] / d['n_matched'] / 100:.1f}%",
                  f"Min inside percentage: {lo / 100:.1f}%",
                  f"Max inside percentage: {hi / 100:.1f}%"]
    return "\n".join(lines)


def run_sharded(frame_items, process_local, device="cpu", group=None):
    """frame_items: the full ascending list of per-frame work items (every rank builds the same
    list); process_local(list_of_items) -> {frame: rows} runs the hot path on this rank's GPU.
    Returns (rows_by_frame over all ranks, reduced vector, min, max)."""
    import torch.distributed as dist
    ws = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    rk = dist.get_rank(group) if ws > 1 else 0
    mine = shard_frames(frame_items, rk, ws)
    local = process_local(mine) if mine else {}
    vec, lo, hi = local_aggregates(local)
    vec, lo, hi = allreduce_aggregates(vec, lo, hi, device, group)
    return gather_rows(local, device, group), vec, lo, hi


def process_frames_distributed(seq=0, cam_id=0, segmenter=None, image_loader=None, kitti360_path=None,
                               master_csv_path="results/master_car_statistics.csv", frames=None,
                               erode_iters=0, v3_pipeline=False, timestamp=None, batch_frames=32):
    """cvs_erosion.process_frames over all ranks of an initialised process group: rank r processes
    frames r, r+W, ... on GPU LOCAL_RANK in batches of ``batch_frames``; rank 0 writes the CSV (same rows, same
    order as one process would) and then, like the reference (cvs_erosion.py:379), prints analyze_master_csv of that
    file -- which covers earlier runs too when the file already existed; the all-reduced aggregates of THIS run are
    returned (format_overall_analysis prints them in the same layout)."""
    import os
    import torch
    import torch.distributed as dist
    from . import pipeline

    root = kitti360_path or os.environ["KITTI360_DATASET"]
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = dist.get_backend() if dist.is_initialized() else "none"
    if backend == "nccl":
        torch.cuda.set_device(local_rank)                   # collectives stage on the current device: one GPU per rank
        device = torch.device("cuda", local_rank)
    else:
        device = "cpu"
    _, camera, velo_to_cam, velo_to_rect, velo = pipeline.sequence_setup(root, seq, cam_id)
    todo = velo.available_frames() if frames is None else list(frames)

    def process_local(my_frames):
        items = pipeline.iter_frame_inputs(root, seq, cam_id, segmenter, image_loader, camera, velo_to_cam, velo, my_frames,
                                           boxes_as_arrays=True)                  # (only the statistics leave this function)
        out = {}
        for batch in pipeline._batches(items, batch_frames):                        # one batch of the shard in memory at a time
            res = pipeline.run_frames(batch, velo_to_rect, camera, 50.0, 10, True, erode_iters, v3_pipeline, local_rank)
            out.update({r["frame"]: r["car_statistics"] for r in res if r["n_valid"] > 0})
        return out

    rows, vec, lo, hi = run_sharded(todo, process_local, device)
    if not dist.is_initialized() or dist.get_rank() == 0:
        for frame, st in rows.items():
            if st:
                pipeline.append_to_master_csv(st, frame, master_csv_path, timestamp)
        pipeline.analyze_master_csv(master_csv_path)
    return rows, vec, lo, hi
