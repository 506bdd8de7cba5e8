"""bench.py's own launcher (no GPU needed): `python bench.py --gpus N` without a launcher around it must start N ranks through
torch.distributed.run from a parent that never touches the GPU, pass its arguments on, and leave with the child's exit code."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(REPO, "bench.py")


def _env(**kw):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(kw)
    return e


def test_gpus_n_builds_the_torchrun_command():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--dist-backend", "gloo",
                          "--force-device", "0", "--dry-launch"], env=_env(), capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    d = json.loads(out.stdout.strip().splitlines()[-1])
    cmd = d["cmd"]
    assert d["would_launch"] == 2 and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "2" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert int(cmd[cmd.index("--master-port") + 1]) > 0 and BENCH in cmd
    tail = cmd[cmd.index(BENCH) + 1:]
    assert tail == ["--gpus", "2", "--steps", "3", "--warmup", "1", "--dist-backend", "gloo", "--force-device", "0"]


def test_failed_ranks_give_a_nonzero_exit():
    """On a box without a GPU the ranks refuse to run; the self-launching parent must report that, not swallow it."""
    import torch
    if torch.cuda.is_available():
        return
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu", "--dist-backend", "gloo",
                          "--force-device", "0"], env=_env(), capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert "needs an MI355X" in (out.stdout + out.stderr)


def test_a_launched_rank_does_not_relaunch():
    """Under a launcher (WORLD_SIZE set) bench.py is a rank: without a GPU it stops with the GPU message, it never spawns."""
    import torch
    if torch.cuda.is_available():
        return
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--no-cpu"], env=_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"),
                         capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "needs an MI355X" in (out.stdout + out.stderr)


def test_pmc_traffic_takes_the_steady_state_step_kernel(tmp_path, monkeypatch):
    """The step kernel shows up in several instantiations in a counter pass (the pipeline's first launch carries only a
    mask pack, the drain's launches only tails and summaries): roofline.traffic quotes the one launched most, and
    nothing at all when the pass ran other kernel sources or another launch size."""
    import json
    import bench
    prof = tmp_path / "profiles"
    prof.mkdir()
    data = {"_meta": {"kernel_source_sha16": "abc", "points_per_launch": 16_000_000},
            "void lpf_step_t<4, 7u, unsigned char, false>": {"launches": 2, "hbm_bytes_per_launch": 38e6},
            "void lpf_step_t<8, 7u, unsigned char, false>": {"launches": 24, "hbm_bytes_per_launch": 516e6},
            "void lpf_step_t<4, 7u, unsigned int, false>": {"launches": 4, "hbm_bytes_per_launch": 9e6},
            "void lpf_k1_project_t<4, 7u, unsigned char>": {"launches": 24, "hbm_bytes_per_launch": 464e6}}
    (prof / bench.PMC_FILE).write_text(json.dumps(data))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernel_source_sha", lambda: "abc")
    assert bench.pmc_traffic(16_000_000, "lpf_step_t")[0] == 516e6
    assert bench.pmc_traffic(16_000_000, "lpf_k1_project_t")[0] == 464e6
    assert bench.pmc_traffic(2_000_000, "lpf_step_t")[0] is None
    monkeypatch.setattr(bench, "kernel_source_sha", lambda: "other")
    got, why = bench.pmc_traffic(16_000_000, "lpf_step_t")
    assert got is None and "re-run" in why
