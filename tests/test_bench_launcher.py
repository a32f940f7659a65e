"""CPU-only: bench.py's rank launcher.  `python bench.py --gpus N` must start its own N ranks
when WORLD_SIZE is unset, refuse a WORLD_SIZE that disagrees with --gpus before any GPU call, and exit
non-zero when a rank fails -- a scaling run must never record one GPU's work under n_gpus = N."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def test_resolve_world_single_rank_without_env():
    assert bench.resolve_world(bench.parse([]), {}) == ("rank", 1)


def test_resolve_world_launches_its_own_ranks():
    assert bench.resolve_world(bench.parse(["--gpus", "4"]), {}) == ("launch", 4)


def test_resolve_world_accepts_a_matching_world():
    assert bench.resolve_world(bench.parse(["--gpus", "8"]), {"WORLD_SIZE": "8"}) == ("rank", 8)


@pytest.mark.parametrize("gpus,ws", [(2, "3"), (1, "2"), (8, "1")])
def test_resolve_world_refuses_a_mismatch(gpus, ws):
    with pytest.raises(SystemExit) as e:
        bench.resolve_world(bench.parse(["--gpus", str(gpus)]), {"WORLD_SIZE": ws})
    assert e.value.code not in (0, None)


def test_launcher_command_is_one_rank_per_gpu_on_localhost():
    cmd = bench.launcher_command(4, ["--gpus", "4", "--steps", "7"], port=29777)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29777"
    assert cmd[-5:] == [os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "7"]


def test_mismatch_exits_non_zero_as_a_process():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "WORLD_SIZE=3" in p.stderr and p.stdout.strip() == ""


def test_launcher_forwards_a_failing_rank(tmp_path):
    """No GPU here: both ranks die at torch.cuda.set_device; the parent must not print a result and must fail."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert p.returncode != 0
    assert '"metric"' not in p.stdout
