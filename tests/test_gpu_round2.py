"""Round-2 GPU checks: bench.py's rank launcher and RCCL path on the one-GPU box, the no-LDS-limit form of
the row-wise sweeps, and the measured streaming ceilings."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from cmtf_pls_amd.engine import default_options
import torch

import oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--shape", "2048", "128", "128", "--steps", "4", "--warmup", "2", "--no-cpu", "--no-ceilings"]      # incl. the fit legs


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
    # the sharded fit legs ran over RCCL too (normal equations / S / norms all-reduced) and agree with each other
    assert len(out["fit"]["n_iter"]) == 10 and out["fit"]["xcov"]["n_iter"] == out["fit"]["n_iter"]
    assert out["fit"]["xcov"]["max_abs_dT_vs_direct"] < 1e-5


def test_bench_launches_two_ranks_by_itself():
    """`python bench.py --gpus 2` with no WORLD_SIZE: the parent spawns its ranks (both on cuda:0 here, gloo
    instead of RCCL, which refuses two ranks on one device) and forwards rank 0's single JSON line."""
    out = _bench(["--gpus", "2"] + SMALL, {"BENCH_ONE_DEVICE": "1", "BENCH_BACKEND": "gloo"})
    assert out["n_gpus"] == 2 and out["config"]["rows_per_gpu"] == 1024
    assert out["collectives"]["ranks"] == 2 and out["collectives"]["backend"] == "gloo"
    assert out["config"]["rccl_ranks"] == 0            # gloo rehearsal: no RCCL ranks claimed
    assert len(out["fit"]["n_iter"]) == 10 and out["fit"]["xcov"]["n_iter"] == out["fit"]["n_iter"]


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
    ref = torch.arange(n, dtype=torch.float32, device="cuda:0")
    a = ref.clone()
    b = torch.zeros_like(a)
    for mp, rb in ((0, 0), (1, 4096), (2, 16384), (2, 65536), (3, 8192), (3, 65536)):
        assert be.ceiling("copy", a, rb, 512, dst=b, map=mp)
        assert torch.equal(a, b)
        b.zero_()
        assert be.ceiling("rmw", a, rb, 512, map=mp)
        assert torch.equal(a, -ref)
        assert be.ceiling("rmw", a, rb, 512, map=mp)
        assert torch.equal(a, ref)
        assert be.ceiling("read", a, rb, 512, map=mp)
    assert not be.ceiling("read", a, 16384 * 16, 512, map=2)       # a map that does not take the row length declines
    torch.cuda.synchronize()


# ---- component epilogue / projection / reconstruction on the device ----------------
@pytest.fixture(scope="module")
def be():
    from cmtf_pls_amd.backend import HipBackend
    return HipBackend("cuda:0")


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


@pytest.mark.parametrize("k", [1, 3, 10, 33, 64])
def test_normal_solve_matches_lstsq_on_badly_scaled_scores(be, k):
    """lstsq(T, u, rcond=-1) (tpls.py:110-112) through the equilibrated normal equations: score columns spread
    over 12 orders of magnitude (late components of a well-explained X) keep their coefficients."""
    rng = np.random.default_rng(k)
    T = rng.normal(size=(500, k)) * np.logspace(0, -12, k)[None, :]
    T[:, 1:] += 0.3 * T[:, :1] * np.logspace(0, -12, k)[None, 1:]          # not orthogonal
    u = rng.normal(size=500)
    want = np.linalg.lstsq(T, u, rcond=-1)[0]
    got = be.normal_solve(_dev(T.T @ T), _dev(T.T @ u)).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=0)


def test_normal_solve_drops_zero_and_dependent_columns(be):
    rng = np.random.default_rng(3)
    T = rng.normal(size=(200, 5))
    T[:, 2] = 0.0                                  # an all-zero score column
    T[:, 4] = 2.0 * T[:, 1]                        # an exactly dependent one
    u = rng.normal(size=200)
    got = be.normal_solve(_dev(T.T @ T), _dev(T.T @ u)).cpu().numpy()
    assert got[2] == 0.0 and got[4] == 0.0 and np.all(np.isfinite(got))
    keep = [0, 1, 3]
    np.testing.assert_allclose(got[keep], np.linalg.lstsq(T[:, keep], u, rcond=None)[0], rtol=1e-10)


def test_projection_fixup_kernels(be):
    rng = np.random.default_rng(8)
    I, R = 1000, 7
    M = rng.normal(size=(I, R))
    U = rng.normal(size=(R, R))
    want = np.linalg.solve((np.eye(R) + np.triu(U, 1)).T, M.T).T
    got = be.unit_upper_solve_rows(_dev(M), _dev(U)).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12)
    # with the centring shift of an uncentred MTTKRP and the missing-value flag
    shift = rng.normal(size=R)
    flag = torch.zeros(1, dtype=torch.int32, device="cuda:0")
    got = be.unit_upper_solve_rows(_dev(M), _dev(U), _dev(shift), flag).cpu().numpy()
    np.testing.assert_allclose(got, np.linalg.solve((np.eye(R) + np.triu(U, 1)).T, (M - shift).T).T, rtol=1e-12, atol=1e-12)
    assert int(flag.item()) == 0
    Mn = M.copy()
    Mn[I - 3, R - 1] = np.nan
    be.unit_upper_solve_rows(_dev(Mn), _dev(U), _dev(shift), flag)
    assert int(flag.item()) == 1
    LA, LB = rng.normal(size=(9, R)), rng.normal(size=(6, R))
    G = torch.empty(R * R, dtype=torch.float64, device="cuda:0")
    be.kr_gram(_dev(LA), G, True)
    be.kr_gram(_dev(LB), G, False)
    np.testing.assert_allclose(G.cpu().numpy().reshape(R, R), (LA.T @ LA) * (LB.T @ LB), rtol=1e-13)
    KR = (LA[:, None, :] * LB[None, :, :]).reshape(-1, R)
    np.testing.assert_array_equal(be.khatri_rao(_dev(LA), _dev(LB)).cpu().numpy(), KR)


def test_kr_axpy_kernel(be):
    rng = np.random.default_rng(19)
    for (A, B, R, k) in [(7, 9, 5, 3), (1, 300, 4, 4), (128, 128, 10, 10), (3, 5, 64, 64), (6, 4, 3, 0)]:
        v, WA, WB, c = rng.normal(size=A * B), rng.normal(size=(A, R)), rng.normal(size=(B, R)), rng.normal(size=max(k, 1))
        want = v - (WA[:, None, :k] * WB[None, :, :k]).reshape(A * B, k) @ c[:k]
        got = be.kr_axpy(_dev(v), A, B, _dev(WA), _dev(WB), k, _dev(c)).cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=1e-13, atol=1e-13)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("case", ["tpls", "coupled", "order4"])
def test_xcov_without_writing_x_on_gpu(dtype, case, monkeypatch):
    """algorithm="xcov" without deflating X (two reads per component, FitRun._finish_xcov_nowrite) against the form that
    deflates in place and against the oracle; the device tensor handed over with copy_X=False ends the fit as centred."""
    opt = {}                                              # EngineOptions fields this test overrides
    from cmtf_pls_amd import ctPLS, tPLS
    td = getattr(torch, dtype)
    R = 4
    if case == "order4":
        x, y, _ = O.import_synthetic((600, 12, 8, 6), 5, R, error=0.2, seed=21)
    else:
        x, y, cp = O.import_synthetic((512, 128, 64), 6, R, error=0.2, seed=21)     # (the CPU oracle fit is what takes the time here)
    Xs = [x]
    if case == "coupled":
        Xs.append(cp.factors[0] @ np.random.default_rng(3).normal(size=(40, R)).T + 0.1 * np.random.default_rng(4).normal(size=(512, 40)))
    if dtype == "float32":
        Xs = [a.astype(np.float32).astype(np.float64) for a in Xs]
    fit = O.fit_ctpls(Xs, y, R) if case == "coupled" else O.fit_tpls(Xs[0], y, R)

    # (this test is about writing X or not: the round-3 form that does not even CENTRE it has its own, tests/test_gpu_round3.py)
    opt["xcov_raw"] = False

    def run(nowrite, keep=None):
        opt["xcov_nowrite"] = nowrite
        if case == "coupled":
            m = ctPLS(R, dtype=dtype, algorithm="xcov", copy_X=keep is None, options=default_options().but(**opt))
            m.fit(Xs if keep is None else keep, y)
            return m, m.factor_T, m.R2Xs[0]
        m = tPLS(R, dtype=dtype, algorithm="xcov", copy_X=keep is None, options=default_options().but(**opt))
        m.fit(Xs[0] if keep is None else keep[0], y)
        return m, m.X_factors[0], m.R2X

    a, Ta, r2a = run(False)
    b, Tb, r2b = run(True)
    assert a.n_iter_ == b.n_iter_
    scale = np.abs(fit.T).max(axis=0)
    tight = 1e-9 if dtype == "float64" else 2e-6          # f32 storage: the deflating form rounds X to f32 once per component
    assert (np.abs(Tb - Ta) / scale).max() < tight
    np.testing.assert_allclose(r2b, r2a, rtol=0, atol=1e-8 if dtype == "float64" else 1e-6)
    np.testing.assert_allclose(b.coef_, a.coef_, rtol=0, atol=(1e-8 if dtype == "float64" else 1e-5) * np.abs(a.coef_).max())
    rtol = 1e-7 if dtype == "float64" else 1e-5
    assert (np.abs(Tb - fit.T) / scale).max() < rtol
    np.testing.assert_allclose(r2b, fit.r2x[0], rtol=0, atol=rtol)
    np.testing.assert_allclose(b.R2Y, fit.r2y, rtol=0, atol=rtol)
    assert list(b.n_iter_) == list(fit.n_iter) or dtype == "float32"
    # copy_X=False: the caller's tensors are centred by the fit and then only read
    dev = [torch.from_numpy(v).to(td).to("cuda:0") for v in Xs]
    c, Tc, _ = run(True, keep=dev)
    np.testing.assert_allclose(Tc, Tb, rtol=0, atol=0)
    for d, v in zip(dev, Xs):
        centred = v - v.mean(axis=0)
        np.testing.assert_allclose(d.cpu().numpy().astype(np.float64), centred, rtol=0, atol=(1e-12 if dtype == "float64" else 1e-6) * np.abs(v).max())


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("coupled", [False, True])
def test_transform_reads_the_callers_tensor_once_and_in_place(dtype, coupled):
    """transform / predict of NaN-free device tensors (tpls.py:122-186; cmtf.py:142-231): no copy of X, no centred
    copy, X untouched -- the MTTKRP runs on the uncentred rows and the centring is applied to its I x R output.  The
    means here are 50x the spread of the data, so a centring lost to cancellation would show at once."""
    from cmtf_pls_amd import ctPLS, tPLS
    td = getattr(torch, dtype)
    x, y, _ = O.import_synthetic((768, 128, 64), 6, 4, error=0.2, seed=12)      # (the CPU oracle fit is what takes the time here)
    x = x + 50.0 * np.random.default_rng(1).normal(size=x.shape[1:])
    xm = np.random.default_rng(2).normal(size=(768, 40)) + 30.0
    if dtype == "float32":
        x, xm = x.astype(np.float32).astype(np.float64), xm.astype(np.float32).astype(np.float64)
    if coupled:
        m = ctPLS(3, dtype=dtype)
        m.fit([x, xm], y)
        fit = O.fit_ctpls([x, xm], y, 3)
    else:
        m = tPLS(3, dtype=dtype)
        m.fit(x, y)
        fit = O.fit_tpls(x, y, 3)
    new = [x[:512], xm[:512]] if coupled else [x[:512]]
    dev = [torch.from_numpy(a).to(td).to("cuda:0") for a in new]
    keep = [d.clone() for d in dev]
    want = O.transform(fit, new)
    arg = dev if coupled else dev[0]
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    got = m.transform(arg)
    peak = torch.cuda.max_memory_allocated() - base
    assert peak < dev[0].numel() * dev[0].element_size() // 4, f"transform allocated {peak} bytes for a {dev[0].numel() * dev[0].element_size()}-byte X"
    assert all(torch.equal(d, k) for d, k in zip(dev, keep))                       # inputs are never modified
    rtol = 1e-5 if dtype == "float32" else 1e-9
    scale = np.abs(want).max(axis=0)
    assert (np.abs(got - want) / scale).max() < rtol
    np.testing.assert_allclose(m.predict(arg), O.predict(fit, new), rtol=0, atol=rtol * 10 * np.abs(y).max())
    # one missing value: the output flags it and the masked sequential path answers, on a private copy
    dev[0][5, 7, 9] = float("nan")
    new[0] = new[0].copy()
    new[0][5, 7, 9] = np.nan
    keep = [d.clone() for d in dev]
    got = m.transform(arg)
    want = O.transform(fit, new)
    assert (np.abs(got - want) / scale).max() < rtol
    assert all(torch.equal(torch.nan_to_num(d), torch.nan_to_num(k)) and torch.equal(torch.isnan(d), torch.isnan(k)) for d, k in zip(dev, keep))


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("shape,R", [((40, 12, 16), 3), ((33, 20), 4), ((17, 6, 4, 8), 2), ((25, 8, 8), 19), ((21, 5, 7), 3), ((19, 13), 2)])
def test_reconstruction_on_device_matches_oracle(shape, R, dtype):
    """X_reconstructed (tpls.py:188-189) = factors_to_tensor (util.py:18-20) + X_mean through cmtfpls_recon_*."""
    from cmtf_pls_amd import tPLS
    x, y, _ = O.import_synthetic(shape, 3, 3, error=0.2, seed=9)
    if dtype == "float32":
        x, y = x.astype(np.float32).astype(np.float64), y.astype(np.float32).astype(np.float64)
    m = tPLS(R, dtype=dtype)
    m.fit(x, y, max_iter=30)
    rec = m.X_reconstructed()
    lit = O.cp_factors_to_tensor(m.X_factors) + m.X_mean                 # the literal reference formula on the fitted factors
    # the host array is the exact float64 reconstruction whatever the storage type (round 3); the device form
    # stays in the storage type (f32 storage: rounded once, 6e-8 relative)
    np.testing.assert_allclose(rec, lit, rtol=1e-12, atol=1e-12 * np.abs(lit).max())
    assert rec.dtype == np.float64 and rec.shape == x.shape
    part = m.X_reconstructed(rows=slice(5, 11), device=True)
    assert part.is_cuda and tuple(part.shape) == (6,) + shape[1:]
    tol = 0.0 if dtype == "float64" else 2e-7
    np.testing.assert_allclose(part.cpu().numpy().astype(np.float64), rec[5:11], rtol=tol, atol=tol * np.abs(lit).max())


def test_more_components_than_latent_factors_noise_free():
    """R above the effective rank on noise-free data.  With 2 latent factors every score lies in a
    2-dimensional sample space: T is exactly rank 2 (singular values 59, 22, 5e-15, 4e-15) and the deflated Y is ~0,
    so the reference's min-norm lstsq(T, u, rcond=-1) returns ~1e-17 for the extra components.  The device solve
    (equilibrated Cholesky that drops dependent columns) must agree: same meaningful coefficients, ~0 elsewhere,
    nothing non-finite."""
    from cmtf_pls_amd import tPLS
    x, y, _ = O.import_synthetic((60, 7, 5), 3, 2, error=0.0, seed=12)
    m = tPLS(4)
    m.fit(x, y, max_iter=50)
    fit = O.fit_tpls(x, y, 4, max_iter=50)
    assert np.all(np.isfinite(m.coef_))
    np.testing.assert_allclose(m.coef_, fit.coef, rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(m.R2Y, fit.r2y, rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(m.X_factors[0][:, :2], fit.T[:, :2], rtol=1e-6, atol=1e-8 * np.abs(fit.T).max())
    assert m.R2Y[-1] > 1 - 1e-10 and fit.r2y[-1] > 1 - 1e-10
    np.testing.assert_allclose(m.predict(x), O.predict(fit, x), rtol=1e-7, atol=1e-9)


def test_fit_epilogue_has_no_host_round_trip():
    """Between start_component and result() the engine reads back ONE status word per iteration through its pinned
    mirror and nothing else: no .cpu(), no .item(), no .tolist()."""
    from cmtf_pls_amd.backend import HipBackend
    from cmtf_pls_amd.engine import NipalsEngine
    x, y, _ = O.import_synthetic((256, 16, 12), 4, 3, error=0.1, seed=2)
    for algorithm in ("direct", "xcov"):
        eng = NipalsEngine(HipBackend("cuda:0"))
        run = eng.begin([_dev(x)], _dev(y), 3, coupled=False, algorithm=algorithm)
        calls = []
        orig = {n: getattr(torch.Tensor, n) for n in ("cpu", "item", "tolist", "numpy")}

        def spy(name):
            def f(self, *a, **k):
                if self.is_cuda:
                    calls.append(name)
                return orig[name](self, *a, **k)
            return f
        try:
            for n in orig:
                setattr(torch.Tensor, n, spy(n))
            for a in range(3):
                run.start_component(a)
                for it in range(30):
                    du = run.iterate(it)
                    if du is not None and du < 1e-8:
                        break
                run.finish_component(a)
        finally:
            for n, f in orig.items():
                setattr(torch.Tensor, n, f)
        assert calls == [], calls
        st = run.result()
        fit = O.fit_tpls(x, y, 3, max_iter=30)
        np.testing.assert_allclose(st.coef, fit.coef, rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(st.r2y, fit.r2y, rtol=1e-8)


def test_copy_x_false_leaves_nothing_for_get_q2y():
    from cmtf_pls_amd import tPLS
    from cmtf_pls_amd.validate import get_q2y
    x, y, _ = O.import_synthetic((40, 5, 4), 2, 2, error=0.1, seed=4)
    m = tPLS(2, copy_X=False)
    m.fit(torch.from_numpy(x).cuda(), y)
    assert m.original_X is None and m.X_miss is None
    with pytest.raises(AssertionError):
        get_q2y(m)


# ---- leave-one-out on the device: all folds in one launch (validate.py:24-33) ---------------
@pytest.mark.parametrize("shape,M,R", [((60, 10, 8), 4, 3), ((40, 8, 10), 1, 2), ((50, 30), 3, 3), ((45, 12, 5), 2, 4)])
def test_loo_all_folds_in_one_launch_matches_literal_refits(shape, M, R):
    """get_q2y through cmtfpls_loo_tpls_f64 == the literal leave-one-out over the oracle (fit on I - 1 samples, predict
    the held-out one), to 1e-8 in Q2Y and 1e-7 per prediction; also == the per-fold refits on the regular engine."""
    from cmtf_pls_amd import tPLS
    from cmtf_pls_amd.validate import get_q2y, loo_predictions
    x, y, _ = O.import_synthetic(shape, M, R, error=0.3, seed=21)
    m = tPLS(R)
    m.fit(x, y)
    pred = loo_predictions(m)
    assert pred is not None and pred.shape == y.shape
    I = shape[0]
    want = np.zeros_like(y)
    for i in range(I):
        keep = np.arange(I) != i
        want[i] = O.predict(O.fit_tpls(x[keep], y[keep], R), x[i:i + 1]).reshape(want[i].shape)
    np.testing.assert_allclose(pred, want, rtol=1e-7, atol=1e-7 * np.abs(y).max())
    q_want = 1 - ((want - y) ** 2).sum() / (y ** 2).sum()
    assert abs(get_q2y(m) - q_want) < 1e-8
    assert abs(get_q2y(m, device_folds=False) - q_want) < 1e-8


def test_loo_declines_shapes_outside_its_form():
    from cmtf_pls_amd import tPLS
    from cmtf_pls_amd.validate import get_q2y, loo_predictions
    x, y, _ = O.import_synthetic((12, 4, 3, 2), 2, 2, error=0.3, seed=22)      # order 4: per-fold refits
    m = tPLS(2)
    m.fit(x, y)
    assert loo_predictions(m) is None
    assert np.isfinite(get_q2y(m))
    x3, y3, _ = O.import_synthetic((14, 5, 4), 2, 2, error=0.3, seed=6)
    x3[2, 1, 1] = np.nan                                                          # missing values: per-fold refits
    m3 = tPLS(2)
    m3.fit(x3, y3)
    assert loo_predictions(m3) is None and np.isfinite(get_q2y(m3))


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("shape,nan", [((60, 12, 16), 0.0), ((50, 12, 16), 0.25), ((40, 24), 0.0), ((30, 6, 4, 8), 0.0), ((30, 5, 7), 0.1)])
def test_literal_r2x_on_device_equals_the_deflation_identity(shape, nan, dtype):
    """calcR2X(X_c, factors_to_tensor(X_factors)) (util.py:7-15, tpls.py:115-117) through cmtfpls_recon_r2_* == the R2X the
    fit books from the deflation sweep == the oracle's literal formula."""
    from cmtf_pls_amd import tPLS
    x, y, _ = O.import_synthetic(shape, 3, 3, error=0.2, seed=19)
    if dtype == "float32":
        x, y = x.astype(np.float32).astype(np.float64), y.astype(np.float32).astype(np.float64)
    if nan:
        x[np.random.default_rng(2).random(x.shape) < nan] = np.nan
    m = tPLS(3, dtype=dtype)
    m.fit(x, y, max_iter=40)
    lit = m.R2X_literal(x)
    tol = 1e-10 if dtype == "float64" else 2e-6
    assert abs(lit - m.R2X[-1]) < tol
    want = O.calc_r2x(x - np.nanmean(x, axis=0), O.cp_factors_to_tensor(m.X_factors))
    assert abs(lit - want) < tol


# ---- randomised end-to-end problems on the HIP path (the CPU suite runs the same cases through the NumPy backend) ----
@pytest.mark.parametrize("seed", range(100, 132))
def test_random_problems_match_oracle_on_gpu(seed):
    """Random orders (2-5), NaNs, 1-D or 2-D Y, tPLS or ctPLS (with an extra matrix block), both algorithms, float64
    storage: HIP engine == oracle (scores, q, coef_, R2, iteration counts, transform of new rows, reconstruction)."""
    from test_engine_logic_cpu import _random_case
    from cmtf_pls_amd import ctPLS, tPLS
    X, Y, R, algorithm, coupled, rng = _random_case(seed)
    if coupled:
        Xm = rng.normal(size=(X.shape[0], int(rng.integers(2, 7))))
        m = ctPLS(R, algorithm=algorithm)
        m.fit([X, Xm], Y)
        fit = O.fit_ctpls([X, Xm], Y, R)
        T, r2x = m.factor_T, m.R2Xs[0]
        new = [X[::2].copy(), Xm[::2].copy()]
        tr, want_tr = m.transform(new), O.transform(fit, new)
        rec = m.Xs_reconstructed()[0]
    else:
        m = tPLS(R, algorithm=algorithm)
        m.fit(X, Y)
        fit = O.fit_tpls(X, Y, R)
        T, r2x = m.X_factors[0], m.R2X
        tr, want_tr = m.transform(X[::2]), O.transform(fit, X[::2])
        rec = m.X_reconstructed()
    if np.isnan(fit.T).any():            # an all-NaN row makes the reference itself produce NaN scores
        assert np.isnan(T).any()
        return
    assert list(m.n_iter_) == list(fit.n_iter)
    s = np.abs(fit.T).max()
    np.testing.assert_allclose(T, fit.T, rtol=1e-6, atol=1e-8 * s)
    np.testing.assert_allclose(m.Y_factors[1], fit.Q, rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(m.coef_, fit.coef, rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(r2x, fit.r2x[0], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(m.R2Y, fit.r2y, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(tr, want_tr, rtol=1e-6, atol=1e-8 * s)
    np.testing.assert_allclose(rec, O.reconstruct(fit, 0), rtol=1e-6, atol=1e-7 * max(1.0, np.nanmax(np.abs(X))))


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("shape,R", [((700, 128, 128), 10), ((300, 24, 32), 5), ((257, 1, 512), 3), ((64, 100, 8), 7), ((90, 6, 256), 2)])
def test_project_rows_kernel_matches_the_sequential_passes(be, shape, R, dtype):
    """cmtfpls_project_rows_*: centring + R masked project-and-deflate steps with the row in registers (one read of the raw
    X) against the passes it replaces (centre, then R x score_deflate), on rows with and without missing values, an
    empty row included: same scores to rounding (same per-element arithmetic and summation order), X untouched."""
    rng = np.random.default_rng(5)
    I, A, B = shape
    td = getattr(torch, dtype)
    x = rng.normal(size=(I, A * B)) + 3.0
    x[rng.random(x.shape) < 0.2] = np.nan
    x[::7] = np.nan_to_num(x[::7], nan=0.5)          # complete rows too
    x[5] = np.nan                                    # an empty row -> NaN scores
    if dtype == "float32":
        x = x.astype(np.float32).astype(np.float64)
    mean = np.nanmean(x, axis=0) + 0.01
    WA, WB = rng.normal(size=(A, R)), rng.normal(size=(B, R))
    WA /= np.linalg.norm(WA, axis=0); WB /= np.linalg.norm(WB, axis=0)
    Xd = _dev(x).to(td)
    keep = Xd.clone()
    got = be.project_rows(Xd, A, B, _dev(WA), _dev(WB), _dev(mean), be.empty(I, R))
    V = 16 // Xd.element_size()
    fits = lambda nt: A * B <= nt * 16 * V and (nt * V) % B == 0    # 16 vectors per lane of a 256- or 1024-thread workgroup
    if not (fits(256) or fits(1024)):                     # longer rows, or B not dividing the workgroup stride: declined
        assert got is None
        return
    assert got is not None
    assert torch.equal(torch.nan_to_num(Xd), torch.nan_to_num(keep)) and torch.equal(torch.isnan(Xd), torch.isnan(keep))
    # the sequential passes on a copy
    Xc = keep.clone()
    rowcnt, _ = be.center(Xc, _dev(mean), True)
    want = be.empty(I, R)
    t = be.empty(I)
    for a in range(R):
        wa, wb = _dev(WA[:, a].copy()), _dev(WB[:, a].copy())
        if be.score_deflate(Xc, A, B, wa, wb, rowcnt, t) is None:
            be.score(Xc, A, B, wa, wb, rowcnt, t)
            be.deflate(Xc, A, B, t, wa, wb)
        want[:, a].copy_(t)
    g, w = got.cpu().numpy(), want.cpu().numpy()
    assert np.array_equal(np.isnan(g), np.isnan(w)) and np.isnan(g[5]).all()
    ok = ~np.isnan(w)
    scale = np.abs(w[ok]).max()
    assert np.abs(g[ok] - w[ok]).max() <= (1e-12 if dtype == "float64" else 1e-6) * scale
    # shapes outside the form are declined, not mangled
    assert be.project_rows(_dev(np.zeros((4, 3 * 5))).to(td), 3, 5, _dev(np.ones((3, 2))), _dev(np.ones((5, 2))), None, be.empty(4, 2)) is None
