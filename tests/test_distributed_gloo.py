"""world_size-2 gloo run of the frame-sharding + aggregate all-reduce (no GPU): each rank gets
its round-robin share of the sample frames, 'processes' them (per-frame rows come from the
golden stats, standing in for the GPU result), and the reduced aggregates / gathered rows must
equal the single-process answer and the text analyze_master_csv prints."""
import contextlib
import io
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import golden_frames, load_golden
from lidar_object_detection_amd import distributed as D
from lidar_object_detection_amd import pipeline

TAG = "rect5_d50"


def _rows_for(frame):
    g = load_golden(frame)
    if "stats_car_id_" + TAG not in g:
        return []
    n = len(g["stats_car_id_" + TAG])
    return [{"car_id": int(g["stats_car_id_" + TAG][i]), "matched_bbox_id": int(g["stats_matched_bbox_id_" + TAG][i]),
             "total_points": int(g["stats_total_points_" + TAG][i]),
             "points_inside_bbox": int(g["stats_points_inside_bbox_" + TAG][i]),
             "points_outside_bbox": int(g["stats_points_outside_bbox_" + TAG][i]),
             "inside_percentage": float(g["stats_inside_percentage_" + TAG][i]),
             "outside_percentage": float(g["stats_outside_percentage_" + TAG][i]), "color": (0, 0, 0)} for i in range(n)]


def _all_frames():
    return [r["frame"] for r in golden_frames()["frames"]]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        seen = []

        def process_local(frames):
            seen.extend(frames)
            return {f: _rows_for(f) for f in frames}

        rows, vec, lo, hi = D.run_sharded(_all_frames(), process_local, "cpu")
        flat = [(f, r["car_id"], r["matched_bbox_id"], r["total_points"], r["points_inside_bbox"], r["inside_percentage"])
                for f, rs in rows.items() for r in rs]
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), vec=vec, lo=lo, hi=hi, seen=np.array(seen),
                 frames=np.array(list(rows.keys())), nrows=np.array([len(v) for v in rows.values()]),
                 flat=np.array(flat, np.float64))
    finally:
        dist.destroy_process_group()


def test_sharded_aggregates_equal_single_process(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    frames = _all_frames()
    single = {f: _rows_for(f) for f in frames}
    vec1, lo1, hi1 = D.local_aggregates(single)
    got = [np.load(str(tmp_path / ("rank%d.npz" % r))) for r in range(world)]
    for r, g in enumerate(got):
        assert g["seen"].tolist() == frames[r::world]                    # round-robin shard
        assert np.array_equal(g["vec"], vec1) and int(g["lo"]) == int(lo1) and int(g["hi"]) == int(hi1)
        assert g["frames"].tolist() == sorted(frames)                    # gathered rows in frame order
        assert g["nrows"].tolist() == [len(single[f]) for f in sorted(frames)]
        want = [(f, r["car_id"], r["matched_bbox_id"], r["total_points"], r["points_inside_bbox"], r["inside_percentage"])
                for f in sorted(frames) for r in single[f]]
        assert np.array_equal(g["flat"], np.array(want, np.float64))     # the padded int64 all-gather carries every row, bit for bit
    # the reduced aggregates print what pandas prints from the CSV
    csv = str(tmp_path / "m.csv")
    with contextlib.redirect_stdout(io.StringIO()) as out:
        for f in sorted(frames):
            pipeline.append_to_master_csv(single[f], f, csv, timestamp="T")
        pipeline.analyze_master_csv(csv)
    ref_lines = [l for l in out.getvalue().splitlines() if ":" in l and not l.startswith(("Created", "Appended"))]
    mine = [l for l in D.format_overall_analysis(vec1, lo1, hi1).splitlines() if ":" in l]
    assert mine == ref_lines
    assert vec1[1] > 40 and vec1[2] > 10


def test_shard_frames_covers_everything_once():
    frames = list(range(20))
    for world in (1, 2, 3, 8):
        parts = [D.shard_frames(frames, r, world) for r in range(world)]
        assert sorted(sum(parts, [])) == frames
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_row_codec_roundtrip_is_bit_exact():
    rows = {100: _rows_for(100), 2717: [], 250: _rows_for(250)}
    rows[100][0]["inside_percentage"] = 100.0 / 3.0                       # not representable in two decimals
    rows[100][0]["color"] = (0.1, 0.2, 0.30000000000000004)
    vec = D.encode_rows(rows)
    back = D.decode_rows(np.concatenate([vec, np.zeros(37, np.int64)]))  # padding is ignored
    assert list(back) == [100, 250, 2717] and back[2717] == []
    for f in rows:
        assert len(back[f]) == len(rows[f])
        for a, b in zip(back[f], rows[f]):
            for k in ("car_id", "matched_bbox_id", "total_points", "points_inside_bbox", "points_outside_bbox",
                      "inside_percentage", "outside_percentage"):
                assert a[k] == b[k]
            assert tuple(a["color"]) == tuple(float(x) for x in b["color"])
