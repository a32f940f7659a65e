"""Round-2 GPU checks: bench.py's rank launcher and RCCL path on the one-GPU box, the no-LDS-limit form of
the row-wise sweeps, and the measured streaming ceilings."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--shape", "2048", "128", "128", "--steps", "4", "--warmup", "2", "--no-cpu", "--no-fit", "--no-ceilings"]


def _bench(extra_args, extra_env):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra_args, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_bench_single_rank_rccl_collectives():
    """BENCH_FORCE_DIST=1: one rank creates its RCCL communicator and the engine issues its two per-iteration
    all-reduces (Z and Y^T t) through it, with segment-wise graph capture around them -- everything of the
    N-GPU path a one-GPU box can run."""
    out = _bench(["--gpus", "1"] + SMALL, {"BENCH_FORCE_DIST": "1"})
    assert out["n_gpus"] == 1 and out["config"]["rccl_ranks"] == 1
    c = out["collectives"]
    assert c["backend"] == "nccl" and c["forced_single_rank"] and abs(c["collectives_per_step"] - 2.0) < 1e-9
    assert str(128 * 128 * 8) in c["allreduce_ms_by_bytes"] and str(16 * 8) in c["allreduce_ms_by_bytes"]
    assert out["value"] > 0 and out["config"]["hip_graphs"]


def test_bench_launches_two_ranks_by_itself():
    """`python bench.py --gpus 2` with no WORLD_SIZE: the parent spawns its ranks (both on cuda:0 here, gloo
    instead of RCCL, which refuses two ranks on one device) and forwards rank 0's single JSON line."""
    out = _bench(["--gpus", "2"] + SMALL, {"BENCH_ONE_DEVICE": "1", "BENCH_BACKEND": "gloo"})
    assert out["n_gpus"] == 2 and out["config"]["rows_per_gpu"] == 1024
    assert out["collectives"]["ranks"] == 2 and out["collectives"]["backend"] == "gloo"
    assert out["config"]["rccl_ranks"] == 0            # gloo rehearsal: no RCCL ranks claimed


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_matrix_block_wider_than_the_lds(dtype):
    """An order-2 X has A = 1, B = P: with P > 12286 the loadings do not fit the 96 KB LDS staging area and the
    sweeps read them through L2 instead (no shape limit, as the reference has none)."""
    from cmtf_pls_amd import tPLS
    rng = np.random.default_rng(5)
    I, P, M, L = 96, 13000, 3, 3
    A0 = rng.normal(size=(I, L))
    x = A0 @ rng.normal(size=(P, L)).T + 0.1 * rng.normal(size=(I, P))
    y = A0 @ rng.normal(size=(M, L)).T + 0.1 * rng.normal(size=(I, M))
    if dtype == "float32":
        x, y = x.astype(np.float32).astype(np.float64), y.astype(np.float32).astype(np.float64)
    for algorithm in ("direct", "xcov"):
        m = tPLS(2, dtype=dtype, algorithm=algorithm)
        m.fit(x, y)
        fit = O.fit_tpls(x, y, 2)
        rtol = 1e-7 if dtype == "float64" else 1e-5
        s = np.abs(fit.T).max()
        np.testing.assert_allclose(m.X_factors[0], fit.T, rtol=rtol, atol=rtol * s)
        np.testing.assert_allclose(m.X_factors[1], fit.loadings[0][0], rtol=rtol, atol=rtol)
        np.testing.assert_allclose(m.R2X, fit.r2x[0], rtol=rtol, atol=rtol)
        np.testing.assert_allclose(m.transform(x[:7]), O.transform(fit, x[:7]), rtol=rtol, atol=rtol * s)
        xt = x[:5].copy()
        xt[2, 100] = np.nan
        np.testing.assert_allclose(m.transform(xt), O.transform(fit, xt), rtol=rtol, atol=rtol * s)


def test_ceiling_kernels_move_the_bytes():
    from cmtf_pls_amd.backend import HipBackend
    be = HipBackend("cuda:0")
    n = 1 << 22
    a = torch.arange(n, dtype=torch.float32, device="cuda:0")
    b = torch.zeros_like(a)
    for rb in (0, 4096):
        be.ceiling("copy", a, rb, 512, dst=b)
        assert torch.equal(a, b)
        b.zero_()
        be.ceiling("rmw", a, rb, 512)
        assert torch.equal(a, -torch.arange(n, dtype=torch.float32, device="cuda:0"))
        be.ceiling("rmw", a, rb, 512)
        assert torch.equal(a, torch.arange(n, dtype=torch.float32, device="cuda:0"))
        be.ceiling("read", a, rb, 512)
    torch.cuda.synchronize()
