"""One rank of tests/test_gpu_distributed.py (not a test module): a fresh process that joins a gloo group and runs
process_frames_distributed -- the real per-rank HIP path -- on the dataset tree the test built from the golden frames.
usage: dist_gpu_worker.py <rank> <world> <port> <root> <csv> <out.npz>     (rank "launch": start <world> ranks and wait)"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def fixture_setup(root):
    """sequence_setup + segmenter for the tree: calibration from the golden arrays (the tree holds scans, images and box files),
    the frames' deterministic masks as the 'segmenter'"""
    import numpy as np
    from conftest import load_calib
    from lidar_object_detection_amd import kitti360, pipeline
    calib = load_calib()
    W, H = int(calib["width"]), int(calib["height"])
    cam = kitti360.CameraPerspective.from_arrays(calib["K"], calib["R_rect"], W, H)
    seq = "2013_05_28_drive_0000_sync"
    velo = kitti360.Kitti360Viewer3DRaw(seq=0, root_dir=root)
    pipeline.sequence_setup = lambda path, s=0, c=0: (seq, cam, calib["TrVeloToCam"], calib["TrVeloToRect"], velo)
    masks_of = dict(np.load(os.path.join(root, "masks.npz")))

    def segmenter(image_path):
        m = np.unpackbits(masks_of[str(int(os.path.basename(image_path).split(".")[0]))], axis=-1)[..., :W].astype(np.float32)
        return None, m, pipeline.default_colors(len(m)), np.zeros((len(m), 4), np.float32), np.ones(len(m))
    return segmenter


def main():
    rank, world, port, root, csv, out = sys.argv[1:7]
    if rank == "launch":                                    # a parent that never touches the GPU starts the ranks
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), str(r), world, port, root, csv, out]) for r in range(int(world))]
        sys.exit(max(p.wait() for p in procs))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=rank, WORLD_SIZE=world, LOCAL_RANK="0")     # both ranks on GPU 0
    import contextlib
    import io
    import numpy as np
    import torch.distributed as dist
    from lidar_object_detection_amd import distributed as D
    segmenter = fixture_setup(root)
    dist.init_process_group("gloo", rank=int(rank), world_size=int(world))
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            rows, vec, lo, hi = D.process_frames_distributed(0, 0, segmenter=segmenter, image_loader=lambda p: p, kitti360_path=root,
                                                             master_csv_path=csv, timestamp="T", batch_frames=3)
        from lidar_object_detection_amd import _native
        lib = _native.load()
        np.savez(out % int(rank), vec=vec, lo=lo, hi=hi, frames=np.array(sorted(rows)), nrows=np.array([len(rows[f]) for f in sorted(rows)]),
                 backend=dist.get_backend(), world=dist.get_world_size(), build_id=lib._lpf_info["build_id"])
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
