"""Sample-mode sharding of the engine with world_size 2 over gloo on CPU (NumPy test backend).

Each rank holds half of the rows of X and Y; the all-reduced quantities (Z, Y^T t, |du|^2, normal
equations, norms, column statistics) must make the sharded fit equal to the single-process fit and
to the oracle: same loadings, q, coef_, R2X, R2Y, iteration counts on both ranks, and the local rows
of T / U.  Also covers the NaN-masked path, whose rescale uses GLOBAL observation counts.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, case, ret):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle as O
        from cmtf_pls_amd import ctPLS, tPLS
        from cmtf_pls_amd.engine import Comm
        from numpy_backend import NumpyBackend

        x, y, cp = O.import_synthetic((60, 8, 6), 70 if case.endswith("_m70") else 3, 3, error=0.1, seed=21)   # _m70: more responses than one S tile
        if case in ("nan", "xcov_nan"):
            x[np.random.default_rng(3).random(x.shape) < 0.25] = np.nan
        if case == "xcov_raw":
            x = x + 4.0                                      # (the uncentred form's corrections are summed over the ranks)
        if world == 2:
            rows = slice(rank * 30, (rank + 1) * 30)
        else:                                                # uneven shards: 13 / 20 / 27 rows on 3 ranks
            cuts = [0, 13, 33, 60]
            rows = slice(cuts[rank], cuts[rank + 1])
        if case in ("coupled", "xcov_coupled"):
            xm = cp.factors[0] @ np.random.default_rng(4).normal(size=(9, 3)).T
            m = ctPLS(3, backend=NumpyBackend(), comm=Comm(), algorithm="xcov" if case == "xcov_coupled" else "direct")
            m.fit([x[rows], xm[rows]], y[rows])
            fit = O.fit_ctpls([x, xm], y, 3)
            T, loads = m.factor_T, m.Xs_factors[0][1:]
            r2x = m.R2Xs[0]
        else:
            m = tPLS(3, backend=NumpyBackend(), comm=Comm(), algorithm="xcov" if case.startswith("xcov") else "direct")
            m.fit(x[rows], y[rows])
            fit = O.fit_tpls(x, y, 3)
            T, loads = m.X_factors[0], m.X_factors[1:]
            r2x = m.R2X
        np.testing.assert_allclose(T, fit.T[rows], rtol=1e-6, atol=1e-8)
        np.testing.assert_allclose(m.Y_factors[0], fit.U[rows], rtol=1e-6, atol=1e-8)
        np.testing.assert_allclose(m.Y_factors[1], fit.Q, rtol=1e-6, atol=1e-8)
        for got, want in zip(loads, fit.loadings[0]):
            np.testing.assert_allclose(np.abs(got), np.abs(want), rtol=1e-6, atol=1e-8)
        np.testing.assert_allclose(m.coef_, fit.coef, rtol=1e-6, atol=1e-8)
        np.testing.assert_allclose(r2x, fit.r2x[0], rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(m.R2Y, fit.r2y, rtol=1e-6, atol=1e-9)
        assert list(m.n_iter_) == list(fit.n_iter)
        # transform of the local rows needs no communication
        if case not in ("coupled", "xcov_coupled"):
            np.testing.assert_allclose(m.transform(x[rows]), fit.T[rows], rtol=1e-6, atol=1e-8)
        ret[rank] = "ok"
    except Exception as e:  # noqa: BLE001
        import traceback
        ret[rank] = traceback.format_exc() + repr(e)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", ["plain", "nan", "coupled", "xcov", "xcov_nan", "xcov_coupled", "xcov_raw", "xcov_m70", "plain_m70"])
def test_world2_matches_oracle(case):
    world = 2
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), case, ret), nprocs=world, join=True)
        assert dict(ret) == {0: "ok", 1: "ok"}, dict(ret)


@pytest.mark.parametrize("case", ["plain", "xcov"])
def test_world3_uneven_shards_match_oracle(case):
    """Three ranks with 13 / 20 / 27 rows: nothing in the engine assumes equal shards (the global sample count and
    every column statistic are all-reduced)."""
    world = 3
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), case, ret), nprocs=world, join=True)
        assert dict(ret) == {0: "ok", 1: "ok", 2: "ok"}, dict(ret)


def _retry_worker(rank, world, port, ret):
    """Forces the rank-1 budget retry in every iteration (sharded direct path): the rejected
    attempt's q must not leak into the convergence norm."""
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle as O
        from cmtf_pls_amd.engine import Comm, NipalsEngine
        from numpy_backend import NumpyBackend

        class Flaky(NumpyBackend):
            def rank1(self, Z, A, B, wA, wB, info=None, n_squarings=None):
                super().rank1(Z, A, B, wA, wB, info=info, n_squarings=n_squarings)
                if n_squarings is not None and n_squarings < self.rank1_squarings:
                    wA.copy_(torch.roll(wA, 1))          # a wrong vector + "not converged"
                    info[0] = 0.0

        x, y, _ = O.import_synthetic((60, 8, 6), 3, 3, error=0.1, seed=21)
        rows = slice(rank * 30, (rank + 1) * 30)
        eng = NipalsEngine(Flaky(), Comm())
        X = torch.from_numpy(x[rows].copy())
        Y = torch.from_numpy(y[rows].copy())
        run = eng.begin([X], Y, 3, coupled=False)
        for a in range(3):
            run.start_component(a)
            for it in range(100):
                run.sq_budget = [5]                      # too small on purpose: every iteration retries
                du = run.iterate(it)
                if du is not None and du < 1e-8:
                    break
            run.finish_component(a)
        st = run.result()
        fit = O.fit_tpls(x, y, 3)
        assert list(st.n_iter) == list(fit.n_iter)
        np.testing.assert_allclose(st.T.numpy(), fit.T[rows], rtol=1e-6, atol=1e-8)
        ret[rank] = "ok"
    except Exception as e:  # noqa: BLE001
        import traceback
        ret[rank] = traceback.format_exc() + repr(e)
    finally:
        dist.destroy_process_group()


def test_rank1_budget_retry_sharded():
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_retry_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
        assert dict(ret) == {0: "ok", 1: "ok"}, dict(ret)


def _giveup_worker(rank, world, port, ret):
    """ONE rank's rank-1 extraction gives up (info = [0, -1], NaN loadings: what the one-launch chain of squarings reports when a
    workgroup never became resident) -- a rank-LOCAL event, unlike everything else an iteration's retry is decided from.  The
    NaN reaches every rank through the all-reduced Y^T t, so all ranks repeat the tail (FitRun._peer_failed) and the collectives
    stay matched: same fit as without the incident, on the fused and the unfused direct iteration."""
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle as O
        from cmtf_pls_amd import tPLS
        from cmtf_pls_amd.engine import Comm
        from numpy_backend import NumpyBackend

        class GivesUp(NumpyBackend):
            """Rank 1 gives up on its 3rd and rank 0 on its 7th extraction, once each (a rank switches its chain off afterwards)."""
            calls = 0
            gave = False

            def rank1(self, Z, A, B, wA, wB, info=None, n_squarings=None):
                super().rank1(Z, A, B, wA, wB, info=info, n_squarings=n_squarings)
                self.calls += 1
                if not self.gave and self.calls == (3 if rank == 1 else 7):
                    wA.fill_(float("nan"))
                    wB.fill_(float("nan"))
                    info[0], info[1] = 0.0, -1.0

            def rank1_chain_gave_up(self):
                self.gave = True

        for M in (3, 70):                                   # 3: Y side fused into the sweeps; 70: the unfused iteration
            x, y, _ = O.import_synthetic((60, 8, 6), M, 3, error=0.1, seed=21)
            rows = slice(rank * 30, (rank + 1) * 30)
            be = GivesUp()
            m = tPLS(3, backend=be, comm=Comm())
            m.fit(x[rows], y[rows])
            assert be.gave and any("switched off" in d for d in m.fit_report_["declined"])
            fit = O.fit_tpls(x, y, 3)
            assert list(m.n_iter_) == list(fit.n_iter)
            np.testing.assert_allclose(m.X_factors[0], fit.T[rows], rtol=1e-6, atol=1e-8)
            np.testing.assert_allclose(m.Y_factors[1], fit.Q, rtol=1e-6, atol=1e-8)
        ret[rank] = "ok"
    except Exception as e:  # noqa: BLE001
        import traceback
        ret[rank] = traceback.format_exc() + repr(e)
    finally:
        dist.destroy_process_group()


def test_one_ranks_rank1_chain_gives_up_in_a_sharded_fit():
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_giveup_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
        assert dict(ret) == {0: "ok", 1: "ok"}, dict(ret)
