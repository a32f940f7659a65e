"""The sharded HIP path on the one-GPU box: two processes, both on cuda:0, sample mode split in
halves, all-reduces through gloo on device tensors (RCCL refuses two ranks on one device; the
multi-GPU RCCL run itself is the driver's).  Everything else is the product path: HIP kernels,
quadratic-form convergence norm, per-segment HIP-graph replay between the collectives.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, case, ret):
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle as O
        from cmtf_pls_amd import tPLS
        from cmtf_pls_amd.backend import HipBackend
        from cmtf_pls_amd.engine import Comm, NipalsEngine

        if case in ("xcov_long_rows", "xcov_coupled"):
            # rows long enough for the one-read form of the cross-covariance loop (scorecontract.hip): its r_a = X_0^T t_a and the
            # raw-mode corrections are summed over the ranks; coupled: the several-blocks iteration under the pipelined loop
            from cmtf_pls_amd import ctPLS
            x, y, cp = O.import_synthetic((512, 32, 64), 4, 3, error=0.2, seed=23)
            x = x + 3.0
            half = 256
            rows = slice(rank * half, (rank + 1) * half)
            calls = {"n": 0}
            orig = HipBackend.score_contract

            def counted(self, *a, **k):
                out = orig(self, *a, **k)
                calls["n"] += out is not None
                return out
            HipBackend.score_contract = counted
            if case == "xcov_coupled":
                xm = cp.factors[0] @ np.random.default_rng(4).normal(size=(40, 3)).T - 1.0
                fit = O.fit_ctpls([xm, x], y, 3)
                m = ctPLS(3, device="cuda:0", comm=Comm(), algorithm="xcov")
                m.fit([xm[rows], x[rows]], y[rows])
                T, r2 = m.factor_T, m.R2Xs[1]
                want_r2 = fit.r2x[1]
            else:
                fit = O.fit_tpls(x, y, 3)
                m = tPLS(3, device="cuda:0", comm=Comm(), algorithm="xcov")
                m.fit(x[rows], y[rows])
                T, r2 = m.X_factors[0], m.R2X
                want_r2 = fit.r2x[0]
            assert calls["n"] == 2                                   # components 0 and 1 (the last needs no down-date)
            s = np.abs(fit.T).max()
            assert list(m.n_iter_) == list(fit.n_iter)
            np.testing.assert_allclose(T, fit.T[rows], rtol=1e-7, atol=1e-7 * s)
            np.testing.assert_allclose(m.Y_factors[1], fit.Q, rtol=1e-7, atol=1e-8)
            np.testing.assert_allclose(r2, want_r2, rtol=1e-7, atol=1e-9)
            np.testing.assert_allclose(m.R2Y, fit.r2y, rtol=1e-7, atol=1e-9)
            ret[rank] = "ok"
            return
        x, y, _ = O.import_synthetic((512, 16, 12), 4, 3, error=0.2, seed=23)
        if case in ("nan", "xcov_nan"):
            x[np.random.default_rng(3).random(x.shape) < 0.25] = np.nan
        half = 256
        rows = slice(rank * half, (rank + 1) * half)
        fit = O.fit_tpls(x, y, 3)
        algorithm = "xcov" if case in ("xcov", "xcov_nan") else "direct"     # xcov_nan: S and S2 rebuilt inside the deflation, summed over ranks
        m = tPLS(3, device="cuda:0", comm=Comm(), algorithm=algorithm, graphs=(case == "graphs"))
        m.fit(x[rows], y[rows])
        s = np.abs(fit.T).max()
        np.testing.assert_allclose(m.X_factors[0], fit.T[rows], rtol=1e-7, atol=1e-7 * s)
        np.testing.assert_allclose(m.Y_factors[1], fit.Q, rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(np.abs(m.X_factors[1]), np.abs(fit.loadings[0][0]), rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(m.coef_, fit.coef, rtol=1e-6, atol=1e-8)
        np.testing.assert_allclose(m.R2X, fit.r2x[0], rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(m.R2Y, fit.r2y, rtol=1e-7, atol=1e-9)
        assert list(m.n_iter_) == list(fit.n_iter)
        np.testing.assert_allclose(m.transform(x[rows]), fit.T[rows], rtol=1e-6, atol=1e-7 * s)
        if case == "graphs":
            # the bench's stepping API with graph replay between the two collectives
            eng = NipalsEngine(HipBackend("cuda:0"), Comm())
            X = torch.from_numpy(np.ascontiguousarray(x[rows])).to("cuda:0", torch.float32)
            Y = torch.from_numpy(np.ascontiguousarray(y[rows])).to("cuda:0")
            run = eng.begin([X], Y, 3, coupled=False)
            run.use_graphs = True
            run.start_component(0)
            dus = [run.iterate(it) for it in range(10)]
            assert run._graph_error is None and run.use_graphs and len(run._graphs) >= 3
            assert all(d is not None and np.isfinite(d) for d in dus[1:]) and dus[-1] < dus[1]
        ret[rank] = "ok"
    except Exception as e:  # noqa: BLE001
        import traceback
        ret[rank] = traceback.format_exc() + repr(e)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", ["plain", "nan", "xcov", "xcov_nan", "graphs", "xcov_long_rows", "xcov_coupled"])
def test_two_ranks_one_gpu(case):
    world = 2
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), case, ret), nprocs=world, join=True)
        assert dict(ret) == {0: "ok", 1: "ok"}, dict(ret)
