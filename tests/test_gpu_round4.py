"""Round 4 on the HIP path: observable path selection, the lifted shape limits, per-sample selection of the projection form,
the conditioning guard of the uncentred xcov form, every branch of the pipelined inner loop.

* `fit_report_` / `projection_report_` name what ran (VERDICT r3 "Next" #2); one test per fallback.
* algorithm="xcov" with more than 64 responses (tpls.py:100-102 has no limit on M): S in response tiles of 64.
* transform / predict of a batch WITH missing values: complete samples keep their one-pass MTTKRP scores, only the
  incomplete ones take the masked sequence (tpls.py:128-142 works sample by sample) -- one block, two and THREE coupled
  blocks (cmtf.py:179-210 takes any number).
* uncentred xcov declined for badly offset data; the speculative branches of the pipelined loop on the GPU;
  the one-launch small fit with more components than rank(X_c).
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
from numpy.testing import assert_allclose

import oracle as O
from cmtf_pls_amd.engine import EngineOptions, default_options

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def api():
    import cmtf_pls_amd
    return cmtf_pls_amd


def _f32(a):
    return a.astype(np.float32).astype(np.float64)


def _normwise(got, want):
    got, want = np.asarray(got, dtype=float), np.asarray(want, dtype=float)
    if got.ndim == 1:
        got, want = got[:, None], want[:, None]
    scale = np.abs(want).max(axis=0, keepdims=True)
    return float((np.abs(got - want) / (np.abs(want) + scale)).max())


# ---- reports ---------------------------------------------------------------------------------------------------------------
def test_fit_report_names_the_forms_that_ran(api):
    x, y, _ = O.import_synthetic((512, 64, 64), 8, 4, error=0.1, seed=3)
    x, y = _f32(x), _f32(y)
    d = api.tPLS(3, dtype="float32")
    d.fit(x, y)
    rep = d.fit_report_
    assert rep["form"] == "regular" and rep["algorithm"] == "direct" and rep["backend"] == "hip"
    assert rep["y_side"] == "fused into the sweeps" and rep["storage"] == ["float32"] and rep["declined"] == [] and not rep["graphs"]
    g = api.tPLS(3, dtype="float32", graphs=True)
    g.fit(x, y)
    assert g.fit_report_["graphs"] is True and g.fit_report_["graph_error"] is None
    xc = api.tPLS(3, dtype="float32", algorithm="xcov")
    xc.fit(x, y)
    rep = xc.fit_report_
    assert rep["algorithm"] == "xcov" and rep["raw"] and rep["one_read"] and rep["pipelined"] and not rep["x_written"]
    assert rep["x_passes_per_component"] == "1 read" and rep["declined"] == []
    assert rep["pipeline"]["iterations"] == sum(xc.n_iter_)
    # a row too short for the one-read kernel: reported, not silent
    xs, ys, _ = O.import_synthetic((256, 16, 16), 4, 3, error=0.1, seed=4)
    s = api.tPLS(3, dtype="float32", algorithm="xcov")
    s.fit(_f32(xs), _f32(ys))
    assert s.fit_report_["one_read"] is False and any("one read per component declined" in w for w in s.fit_report_["declined"])
    # missing values: rebuilds S per component, deflation inside the rebuild
    xn = x.copy()
    xn[np.random.default_rng(5).random(x.shape) < 0.2] = np.nan
    n = api.tPLS(3, dtype="float32", algorithm="xcov")
    n.fit(xn, y)
    rep = n.fit_report_
    assert rep["missing"] == [True] and not rep["raw"] and "deflation inside the rebuild of S" in rep["x_passes_per_component"]


@pytest.mark.small_fit
def test_fit_report_of_the_one_launch_small_fit(api):
    x, y, _ = O.import_synthetic((200, 10, 8), 4, 3, error=0.1)
    m = api.tPLS(3, algorithm="xcov", graphs=True)               # neither applies to the one-launch form: the report says so
    m.fit(x, y)
    assert m.fit_report_["form"] == "small_fit" and m.fit_report_["launches"] == 1 and "do not apply" in m.fit_report_["note"]
    r = api.tPLS(3, options=EngineOptions(small_fit=False))
    r.fit(x, y)
    assert r.fit_report_["form"] == "regular"


# ---- more than 64 responses --------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", ["tpls_f32", "tpls_f64", "coupled", "nan"])
def test_xcov_with_96_responses(api, case):
    """The reference takes any number of responses (tpls.py:100-102).  M = 96: S = Y^T X_(0) is built in two response tiles
    (cmtfpls_xcov_*), the inner loop runs on S through the general launches; equal to the direct loop and the oracle."""
    rng = np.random.default_rng(21)
    I, M, R = 384, 96, 3
    lat = rng.normal(size=(I, 5))
    x = np.einsum("ir,jr,kr->ijk", lat, rng.normal(size=(32, 5)), rng.normal(size=(64, 5))) + 0.1 * rng.normal(size=(I, 32, 64))
    y = lat @ rng.normal(size=(5, M)) + 0.1 * rng.normal(size=(I, M))
    x, y = _f32(x), _f32(y)
    dtype = "float64" if case == "tpls_f64" else "float32"
    if case == "nan":
        x[rng.random(x.shape) < 0.2] = np.nan
    blocks = [x, _f32(lat @ rng.normal(size=(5, 48)) + 0.1 * rng.normal(size=(I, 48)))] if case == "coupled" else [x]
    coupled = case == "coupled"
    make = lambda alg: (api.ctPLS if coupled else api.tPLS)(R, dtype=dtype, algorithm=alg)
    a, d = make("xcov"), make("direct")
    for m in (a, d):
        m.fit(blocks if coupled else x, y)
    rep = a.fit_report_
    assert rep["algorithm"] == "xcov" and rep["responses"] == M and "2 response tiles" in rep["s_build"]
    assert any("more than 64 responses" in w for w in rep["declined"])            # (the pipelined single-call iteration is M <= 64)
    assert d.fit_report_["y_side"] == "separate launches"
    Ta, Td = (a.factor_T, d.factor_T) if coupled else (a.X_factors[0], d.X_factors[0])
    tol = 1e-9 if dtype == "float64" else 1e-5
    assert list(a.n_iter_) == list(d.n_iter_)
    assert _normwise(Ta, Td) <= tol
    fit = O.fit_ctpls(blocks, y, R) if coupled else O.fit_tpls(x, y, R)
    assert list(a.n_iter_) == list(fit.n_iter)
    assert _normwise(Ta, fit.T) <= tol and _normwise(a.Y_factors[1], fit.Q) <= tol
    assert_allclose(a.R2Y, fit.r2y, rtol=0, atol=tol)


def test_xcov_kernel_response_tiles_equal_one_gemm(api):
    """cmtfpls_xcov_{f32,f64} and cmtfpls_xcov_ssq_* at M = 64, 65, 96, 130 against torch's f64 matmul."""
    from cmtf_pls_amd.backend import HipBackend
    be = HipBackend("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(3)
    for dtype in (torch.float32, torch.float64):
        X = torch.randn(700, 24 * 32, generator=g, dtype=torch.float64).to(dtype).cuda()
        for M in (64, 65, 96, 130):
            Y = torch.randn(700, M, generator=g, dtype=torch.float64).cuda()
            want = Y.T @ X.double()
            S = be.xcov(X, Y, False)
            assert float((S - want).abs().max()) <= 1e-11 * float(want.abs().max()) + 1e-12, (dtype, M)
            mean = X.double().mean(dim=0)
            S2, ssq = be.xcov_ssq(X, Y, mean, out=be.empty(M, X.shape[1]))
            assert torch.equal(S2, S)
            assert abs(float(ssq.item()) - float(((X.double() - mean) ** 2).sum())) <= 1e-10 * float(ssq.item())
            Xn = X.clone()
            Xn[::7, ::5] = float("nan")
            Sm = be.xcov(Xn, Y, True)
            wantm = Y.T @ torch.nan_to_num(Xn.double(), nan=0.0)
            assert float((Sm - wantm).abs().max()) <= 1e-11 * float(wantm.abs().max()) + 1e-12, (dtype, M)
            if dtype == torch.float32:                                   # the opt-in f32-MFMA form takes the same tiles
                Sx = be.xcov(X, Y, False, mixed=True)
                assert float((Sx - want).abs().max()) <= 2e-5 * float(want.abs().max()), (M, "mixed")


# ---- projection: per-sample form ---------------------------------------------------------------------------------------------
def _coupled_data(n_blocks, I=600, seed=8):
    rng = np.random.default_rng(seed)
    lat = rng.normal(size=(I, 4))
    x = _f32(np.einsum("ir,jr,kr->ijk", lat, rng.normal(size=(32, 4)), rng.normal(size=(64, 4))) + 0.1 * rng.normal(size=(I, 32, 64)))
    blocks = [x]
    if n_blocks >= 2:
        blocks.append(_f32(lat @ rng.normal(size=(4, 512)) + 0.1 * rng.normal(size=(I, 512))))
    if n_blocks >= 3:
        blocks.append(_f32(np.einsum("ir,jr,kr->ijk", lat, rng.normal(size=(8, 4)), rng.normal(size=(16, 4))) + 0.1 * rng.normal(size=(I, 8, 16))))
    y = _f32(lat @ rng.normal(size=(4, 6)) + 0.1 * rng.normal(size=(I, 6)))
    return blocks, y


@pytest.mark.parametrize("n_blocks", [1, 2, 3])
@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_transform_of_a_partly_incomplete_batch(api, n_blocks, dtype):
    """30 % of the SAMPLES have missing values.  The complete samples get exactly the scores they get when transformed on
    their own (one-pass MTTKRP: a sample's result does not depend on the rest of the batch), the incomplete ones the
    reference's masked sequence (oracle); one block and two coupled blocks in registers, three through compact copies."""
    blocks, y = _coupled_data(n_blocks)
    coupled = n_blocks > 1
    R = 5
    m = (api.ctPLS if coupled else api.tPLS)(R, dtype=dtype)
    m.fit(blocks if coupled else blocks[0], y, max_iter=40)
    fit = (O.fit_ctpls(blocks, y, R, max_iter=40) if coupled else O.fit_tpls(blocks[0], y, R, max_iter=40))
    rng = np.random.default_rng(9)
    new = [b[:400].copy() for b in blocks]
    bad = rng.random(400) < 0.3
    for b in new:
        hole = rng.random(b.shape) < 0.25
        hole[~bad] = False
        b[hole] = np.nan
    new[0][np.flatnonzero(bad)[0]] = np.nan                     # an EMPTY row in the first block: NaN score (missingvals.py:37)
    arg = new if coupled else new[0]
    got = m.transform(arg)
    rep = m.projection_report_
    assert rep["incomplete_rows"] == int(bad.sum())
    if n_blocks <= 2:
        assert rep["form"].startswith("one-pass MTTKRP for the complete samples + masked sequence in registers"), rep
    else:
        assert rep["form"].startswith("one-pass MTTKRP for the complete samples + sequential passes on copies"), rep
    want = O.transform(fit, arg)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    ok = ~np.isnan(want).any(axis=1)
    tol = 2e-5 if dtype == "float32" else 1e-8
    assert _normwise(got[ok], want[ok]) <= tol
    alone = m.transform([b[~bad] for b in new] if coupled else new[0][~bad])
    assert m.projection_report_["form"].startswith("one-pass MTTKRP (one read")
    assert np.array_equal(got[~bad], alone)                      # bit for bit: the complete samples never saw the masked path
    # the old behaviour (every row of such a batch through the masked sequence) stays selectable and agrees
    every = (api.ctPLS if coupled else api.tPLS)(R, dtype=dtype, options=default_options().but(project_split_rows=False))
    every.fit(blocks if coupled else blocks[0], y, max_iter=40)
    got2 = every.transform(arg)
    assert np.array_equal(np.isnan(got2), np.isnan(want)) and _normwise(got2[ok], want[ok]) <= tol
    p = m.predict(arg)
    wantp = O.predict(fit, arg)
    assert _normwise(p[ok], wantp[ok]) <= 10 * tol


def test_transform_of_a_mostly_incomplete_batch_skips_the_mttkrp_attempt(api):
    blocks, y = _coupled_data(1)
    m = api.tPLS(4, dtype="float32")
    m.fit(blocks[0], y, max_iter=30)
    new = blocks[0][:300].copy()
    new[np.random.default_rng(3).random(new.shape) < 0.3] = np.nan             # every sample has holes (BASELINE configs[3])
    got = m.transform(new)
    rep = m.projection_report_
    assert rep["probe_incomplete_fraction"] == 1.0 and rep["form"].startswith("masked sequence, every row in registers"), rep
    want = O.transform(O.fit_tpls(blocks[0], y, 4, max_iter=30), new)
    assert _normwise(got, want) <= 2e-5


# ---- ADVICE r3 -------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("offset,raw", [(20.0, True), (1e6, False)])
def test_uncentred_xcov_is_declined_for_badly_offset_data(api, offset, raw):
    """xcov on the uncentred tensor works by cancellation (error ~ 1e-16 |mean| / spread): taken at 20 x the spread, declined
    at 1e6 x (`EngineOptions.xcov_raw_max_offset` = 1e4) in favour of the centred private copy -- which matches the oracle as
    a centred fit does; forcing the raw form there loses the digits the guard protects."""
    x, y, _ = O.import_synthetic((300, 16, 64), 5, 4, error=0.1, seed=12)
    x = x + offset
    m = api.tPLS(4, algorithm="xcov")
    m.fit(x, y)
    rep = m.fit_report_
    assert rep["raw"] is raw
    assert any("uncentred xcov form declined: max|column mean|" in w for w in rep["declined"]) == (not raw)
    fit = O.fit_tpls(x, y, 4)
    assert list(m.n_iter_) == list(fit.n_iter)
    err = _normwise(m.X_factors[0], fit.T)
    assert err <= 1e-8, err
    assert_allclose(m.R2X, fit.r2x[0], rtol=0, atol=1e-9)
    if not raw:
        forced = api.tPLS(4, algorithm="xcov", options=default_options().but(xcov_raw_max_offset=float("inf")))
        forced.fit(x, y)
        assert forced.fit_report_["raw"] is True
        print(f"offset 1e6: centred copy {err:.2e}, forced uncentred form {_normwise(forced.X_factors[0], fit.T):.2e} (normwise T vs oracle)")


def test_every_branch_of_the_pipelined_inner_loop_runs_on_the_gpu(api):
    """FitRun._inner_loop_xcov_pipelined on the device, every branch forced and counted (`fit_report_["pipeline"]`), each time bit-equal
    to the loop that waits after every iteration, the caller's X untouched:
      redone  -- a squaring budget too small for the rank-1 extraction: the tail of the iteration is re-enqueued while a
                 speculative iteration built on the unfinished loadings is already in flight and must be discarded;
      unused  -- the loop converges although the norms did not predict it: the iteration enqueued ahead ran for nothing;
      waited  -- the norms predicted convergence, it did not come: no speculation, the GPU idles through one round trip."""
    from cmtf_pls_amd.backend import HipBackend
    from cmtf_pls_amd.engine import NipalsEngine
    x, y, _ = O.import_synthetic((1024, 32, 64), 6, 10, error=0.3, seed=17)
    X = torch.from_numpy(_f32(x)).float().cuda()
    Y = torch.from_numpy(_f32(y)).cuda()
    X0 = X.clone()
    be = HipBackend("cuda:0")

    def fit(pipeline, tol, max_iter=100, budget=None, R=3):
        eng = NipalsEngine(be, None, EngineOptions(small_fit=False, xcov_pipeline=pipeline))
        run = eng.begin([X], Y.clone(), R, False, algorithm="xcov", owned=[False])
        run.tol = tol
        dus = []
        for a in range(R):
            run.start_component(a)
            if budget is not None:
                run.sq_budget = [budget]
            if pipeline:
                run.inner_loop(a, max_iter, tol)
            else:
                for it in range(max_iter):
                    du = run.iterate(it)
                    dus.append((a, it, du))
                    if du is not None and du < tol:
                        break
            run.finish_component(a)
        st = run.result()
        return st, dus

    def same(a, b):
        assert a.n_iter == b.n_iter
        assert torch.equal(a.T, b.T) and torch.equal(a.Q, b.Q) and torch.equal(a.blocks[0].loadings[0], b.blocks[0].loadings[0])

    # redone: one squaring (+ 4 spare on a component's first iteration) cannot separate sigma_2 / sigma_1 ~ 0.9
    wait, dus = fit(False, 1e-8, budget=1)
    pipe, _ = fit(True, 1e-8, budget=1)
    same(pipe, wait)
    stats = pipe.report["pipeline"]
    assert stats["redone"] > 0 and stats["ahead"] > 0, stats
    # unused: a tolerance every second norm meets (no two norms yet to predict from): the iteration enqueued ahead is discarded
    wait_u, _ = fit(False, 1e30)
    pipe_u, _ = fit(True, 1e30)
    same(pipe_u, wait_u)
    assert pipe_u.report["pipeline"]["unused"] > 0 and pipe_u.n_iter == [2, 2, 2], (pipe_u.report["pipeline"], pipe_u.n_iter)
    # waited: a tolerance between the predicted norm d_{k-1}^2 / d_{k-2} and the norm d_k that actually came (the decay slows down)
    seq = [du for (a, it, du) in dus if a == 0 and du is not None]
    ks = [k for k in range(2, len(seq)) if seq[k - 1] ** 2 / seq[k - 2] < 0.9 * seq[k]]
    assert ks, ("the first component's norms never decay slower than predicted: pick other data", seq[:12])
    k = ks[0]
    tol = float(np.sqrt((seq[k - 1] ** 2 / seq[k - 2]) * seq[k]))
    wait_w, _ = fit(False, tol, R=1)
    pipe_w, _ = fit(True, tol, R=1)
    same(pipe_w, wait_w)
    assert pipe_w.report["pipeline"]["waited"] > 0, (pipe_w.report["pipeline"], tol, seq[:k + 2])
    assert torch.equal(X, X0)                                    # owned=[False]: the caller's tensor is never written


@pytest.mark.small_fit
def test_more_components_than_the_rank_of_the_data_on_both_small_fit_paths(api, small_fit_mode):
    """R exceeds the rank of the centred data (the two paths of a small float64 fit stopped after different
    iteration counts there).  Noise-free CP data of rank 2: after two components the deflated Y is rounding noise, and the
    start vector u = Y[:, 0] of every later component (tpls.py:78) is that noise -- in the reference as here, so their
    loadings, iteration counts and R2X are not comparable quantities.  Pinned against the oracle on BOTH paths: the two
    defined components (factors, iteration counts, R2X), R2Y of all four (Y is exhausted: 1 to rounding), predictions; and
    the noise components stay finite with R2X non-decreasing and <= 1."""
    x, y, _ = O.import_synthetic((40, 6, 5), 3, 2, error=0.0, seed=3)          # X and Y have CP rank 2
    R = 4
    m = api.tPLS(R, options=small_fit_mode)
    m.fit(x, y)
    assert m.fit_report_["form"] == ("small_fit" if small_fit_mode.small_fit else "regular")
    fit = O.fit_tpls(x, y, R)
    assert list(m.n_iter_[:2]) == list(fit.n_iter[:2])
    assert _normwise(m.X_factors[0][:, :2], fit.T[:, :2]) <= 1e-9
    assert_allclose(m.R2X[:2], fit.r2x[0][:2], rtol=0, atol=1e-9)
    assert np.all(np.diff(m.R2X) >= -1e-9) and m.R2X[-1] <= 1 + 1e-9
    assert_allclose(m.R2Y, fit.r2y, rtol=0, atol=1e-9)
    assert np.all(np.isfinite(m.coef_)) and np.all(np.isfinite(m.X_factors[0]))
    assert_allclose(m.predict(x[:7]), O.predict(fit, x[:7]), rtol=1e-7, atol=1e-9)


# ---- multi-GPU readiness a one-GPU box can prove (VERDICT r3 "Next" #3) ----------------------------------------------------------
def _clean_env(extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def _json_line(cmd, env):
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_allreduces_inside_the_iteration_graph_on_one_rccl_rank():
    """EngineOptions.capture_collectives on the one-rank RCCL communicator: either the two all-reduces are captured into the
    iteration's graph (then: bit-identical to the segment-wise form, fewer graphs, overhead reported) or the capture fails and
    the engine says so and keeps the segment-wise form -- never a wrong or lost iteration."""
    out = _json_line([sys.executable, os.path.join(ROOT, "tools", "collectives_in_graph.py"), "2048", "128", "128", "--steps", "40"],
                     _clean_env({}))
    print(json.dumps(out))
    assert out["rccl_ranks"] == 1
    assert out["bit_identical_segment_vs_captured"] and out["bit_identical_sharded_vs_unsharded"]
    cap = out["collectives_in_graph"]
    assert cap["graphs"] and cap["graph_error"] is None
    if cap["collectives_in_graph"]:
        assert cap["n_graphs"] < out["segment_wise"]["n_graphs"]
    else:
        assert any("not captured" in n for n in cap["notes"])


def test_bench_sharded_north_star_line_prints_with_two_ranks():
    """`bench.py --gpus 2 --shape <rows> 256 256 --responses 32` (BASELINE configs[4]'s trailing shape), two gloo ranks on
    cuda:0: the sharded line with rows_per_gpu, the per-rank timings, the collectives and both fit legs is known to print."""
    out = _json_line([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--shape", "1024", "256", "256", "--responses", "32",
                      "--steps", "4", "--warmup", "2", "--no-cpu", "--no-ceilings", "--capture-collectives", "1"],
                     _clean_env({"BENCH_ONE_DEVICE": "1", "BENCH_BACKEND": "gloo"}))
    assert out["n_gpus"] == 2 and out["config"]["rows_per_gpu"] == 512 and out["scaling"] == "strong"
    assert out["collectives"]["ranks"] == 2 and abs(out["collectives"]["collectives_per_step"] - 2.0) < 1e-9
    assert str(256 * 256 * 8) in out["collectives"]["allreduce_ms_by_bytes"] and str(32 * 8) in out["collectives"]["allreduce_ms_by_bytes"]
    # gloo cannot be captured into a HIP graph: the engine must have fallen back to the segment-wise form and said so
    assert out["collectives"]["in_graph"] is False and any("not captured" in n for n in out["collectives"]["notes"])
    assert [r["rank"] for r in out["per_rank"]] == [0, 1]
    for r in out["per_rank"]:
        assert r["rows"] == 512 and r["contraction_ms"] > 0 and r["score_ms"] > 0 and r["rank1_ms"] > 0 and r["allreduce_ms_per_step"] > 0
    assert len(out["fit"]["n_iter"]) == 10 and out["fit"]["xcov"]["n_iter"] == out["fit"]["n_iter"]
    assert out["fit"]["path"]["xcov"]["algorithm"] == "xcov" and out["fit"]["path"]["xcov"]["sharded"] and out["fit"]["path"]["xcov"]["world"] == 2


# ---- leave-one-out beyond the LDS-resident shapes (VERDICT r3 "Next" #5) -------------------------------------------
def test_loo_xcov_kernel_equals_the_lds_kernel_where_both_apply():
    """cmtfpls_loo_xcov_f64 (the fold's loop on its cross-covariance, Gram squarings on the matrix cores) against
    cmtfpls_loo_tpls_f64 (the literal loop with the fold's vectors in LDS) on shapes both take: predictions of every held-out
    sample to 1e-9, equal iteration counts."""
    from cmtf_pls_amd.backend import HipBackend
    be = HipBackend("cuda:0")
    for shape, M, R in (((50, 16, 12), 3, 3), ((40, 9, 40), 5, 4), ((64, 33, 20), 2, 2), ((45, 30), 4, 3)):
        x, y, _ = O.import_synthetic(shape, M, max(R, 3), error=0.2, seed=31)
        X2 = torch.from_numpy(x.reshape(shape[0], -1)).cuda()
        Y = torch.from_numpy(y.reshape(shape[0], -1)).cuda()
        A, B = (1, shape[1]) if len(shape) == 2 else (shape[1], shape[2])
        lds = be.loo_tpls(X2, Y, A, B, R, 1e-8, 100, forms=("lds",))
        xc = be.loo_tpls(X2, Y, A, B, R, 1e-8, 100, forms=("xcov",))
        assert lds is not None and xc is not None and lds[2] == "lds" and xc[2] == "xcov", shape
        assert torch.equal(lds[1], xc[1]), (shape, lds[1].sum().item(), xc[1].sum().item())
        err = float((lds[0] - xc[0]).abs().max() / lds[0].abs().max())
        assert err <= 1e-9, (shape, err)


@pytest.mark.parametrize("shape,M,R", [((48, 80, 96), 3, 3), ((40, 96, 70), 4, 2), ((36, 128, 128), 16, 4), ((20, 256, 256), 32, 2),
                                       ((44, 72, 80), 3, 20),      # more than 16 components: the loadings of all components live in the fold's workspace
                                       ((36, 40, 30), 96, 2), ((30, 80, 72), 100, 3)])      # more than 64 responses (the LDS form declines, the xcov form takes them)
def test_q2y_beyond_the_lds_shapes_equals_literal_refits(api, shape, M, R):
    """validate.get_q2y (validate.py:24-37) at trailing shapes with min(J, K) > 64 -- one refit per fold on the regular engine in
    round 3 -- through cmtfpls_loo_xcov_f64: Q2Y and every held-out prediction equal the literal refits' at 1e-8."""
    from cmtf_pls_amd.validate import get_q2y, loo_predictions
    x, y, _ = O.import_synthetic(shape, M, R + 1, error=0.3, seed=5)
    m = api.tPLS(R)
    m.fit(x, y)
    q_dev = get_q2y(m)
    assert "cmtfpls_loo_xcov_f64" in m.q2y_report_["form"], m.q2y_report_
    pred = loo_predictions(m)
    q_ref = get_q2y(m, device_folds=False)
    assert m.q2y_report_["form"].startswith("one refit per fold")
    assert abs(q_dev - q_ref) <= 1e-8 * max(1.0, abs(q_ref)), (q_dev, q_ref)
    # the held-out predictions themselves, against literal refits of a few folds
    for i in (0, shape[0] // 2, shape[0] - 1):
        keep = np.ones(shape[0], dtype=bool)
        keep[i] = False
        r = api.tPLS(R)
        r.fit(x[keep], y[keep])
        want = r.predict(x[i:i + 1]).reshape(-1)
        assert np.abs(pred[i].reshape(-1) - want).max() <= 1e-8 * max(1.0, np.abs(want).max()), (i, pred[i], want)


# ---- the rank-1 chain of squarings in one launch: what happens when its workgroups are not all resident ----------------------------
def test_rank1_chain_gives_up_loudly_and_the_engine_falls_back_to_launches(api):
    """syrk_chain_kernel needs every workgroup resident (true on a GPU the process owns).  With one row of workgroups missing
    (cmtfpls_rank1_chain_enable(2), a test mode) the partners wait a bounded time, the extraction returns info = [0, -1] with
    NaN loadings -- never a wrong 'converged' -- and the engine switches the chain off for the process, repeats the iteration
    through the launch-per-squaring form and says so in the report: same fit as with the chain off from the start."""
    from cmtf_pls_amd.backend import HipBackend
    be = HipBackend("cuda:0")
    lib = be.lib
    x, y, _ = O.import_synthetic((300, 64, 48), 5, 4, error=0.2, seed=9)
    try:
        lib.cmtfpls_rank1_chain_enable(0)
        ref = api.tPLS(3, algorithm="xcov", options=default_options().but(small_fit=False))
        ref.fit(x, y)
        assert ref.fit_report_["rank1_one_launch_chain"] is False
        lib.cmtfpls_rank1_chain_enable(1)
        on = api.tPLS(3, algorithm="xcov", backend=HipBackend("cuda:0"), options=default_options().but(small_fit=False))
        on.fit(x, y)
        assert on.fit_report_["rank1_one_launch_chain"] is True and on.n_iter_ == ref.n_iter_
        assert np.array_equal(on.X_factors[0], ref.X_factors[0])            # the chain and the launches: the same bits
        # kernel level
        lib.cmtfpls_rank1_chain_enable(2)
        Z = torch.randn(64 * 48, dtype=torch.float64, device="cuda:0")
        wA, wB, info = be.empty(64), be.empty(48), be.zeros(2)
        be.rank1(Z, 64, 48, wA, wB, info=info)
        assert info.cpu().tolist() == [0.0, -1.0] and bool(torch.isnan(wA).any())
        # engine level: direct and xcov
        for algorithm in ("xcov", "direct"):
            lib.cmtfpls_rank1_chain_enable(2)
            be2 = HipBackend("cuda:0")
            m = api.tPLS(3, algorithm=algorithm, backend=be2, options=default_options().but(small_fit=False))
            m.fit(x, y)
            assert lib.cmtfpls_rank1_chain_enabled() == 0
            assert any("switched off" in d for d in m.fit_report_["declined"]), m.fit_report_["declined"]
            assert m.fit_report_["rank1_one_launch_chain"] is False
            assert m.n_iter_ == ref.n_iter_
            assert _normwise(m.X_factors[0], ref.X_factors[0]) <= 1e-9
    finally:
        lib.cmtfpls_rank1_chain_enable(1)


# ---- the column statistics out of the read that builds S (second session of round 4) --------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("I,P,M", [(4096, 16384, 16), (1000, 4096, 32), (777, 1000, 5), (130, 514, 48), (64, 256, 1), (333, 65536, 64)])
def test_xcov_stats_kernel_gives_s_and_the_column_statistics_from_one_read(dtype, I, P, M):
    """cmtfpls_xcov_stats_*: S bit for bit the matrix-core pass of cmtfpls_xcov_*, the column sums and sums of squares of tpls.py:61-71
    from the same read; a missing value shows as a NaN in its column's sum."""
    from cmtf_pls_amd.backend import HipBackend
    be = HipBackend("cuda:0")
    rng = np.random.default_rng(I + P + M)
    x = rng.normal(size=(I, P)) * rng.uniform(0.5, 2.0, size=P) + rng.normal(size=P) * 5.0
    if dtype == torch.float32:
        x = x.astype(np.float32).astype(np.float64)
    y = rng.normal(size=(I, M))
    X, Y = torch.from_numpy(x).cuda().to(dtype), torch.from_numpy(y).cuda()
    S, stats = be.xcov_stats(X, Y, out=be.empty(M, P))
    assert torch.equal(S, be.xcov(X, Y, False, out=be.empty(M, P)))
    got = stats.cpu().numpy()
    assert np.abs(got[:P] - x.sum(axis=0)).max() <= 1e-12 * np.abs(x).sum(axis=0).max()
    assert np.abs(got[P:] - (x * x).sum(axis=0)).max() <= 1e-12 * (x * x).sum(axis=0).max()
    X[I // 2, P // 3] = float("nan")
    _, stats = be.xcov_stats(X, Y, out=be.empty(M, P))
    bad = torch.isnan(stats[:P]).nonzero().reshape(-1).tolist()
    assert bad == [P // 3]
    assert be.xcov_stats(X, torch.zeros(I, 65, dtype=torch.float64, device="cuda:0"), out=be.empty(65, P)) is None


@pytest.mark.parametrize("dtype,shape", [("float32", (400, 128, 128)), ("float64", (300, 24, 32)), ("float32", (350, 96)), ("float32", (200, 256, 256))])
def test_xcov_raw_fit_reads_x_once_before_its_first_component(api, monkeypatch, dtype, shape):
    """tpls.py:61-71 + the first S from ONE read of the caller's uncentred tensor (backend.xcov_stats): no statistics pass over X, no
    xcov_ssq; same fit as with the statistics pass first (EngineOptions.xcov_stats_with_s = False) -- the means are summed in another
    order, so equal to rounding, not bit for bit -- and as the oracle; the tensor is not written."""
    from cmtf_pls_amd.backend import HipBackend
    x, y, _ = O.import_synthetic(shape, 5, 4, error=0.1, seed=41)
    x = x + 6.0
    if dtype == "float32":
        x, y = _f32(x), _f32(y)
    P = int(np.prod(shape[1:]))
    seen = {"colstats_x": 0, "xcov_stats": 0, "xcov_ssq": 0}
    for name in ("colstats", "xcov_stats", "xcov_ssq"):
        orig = getattr(HipBackend, name)

        def counted(self, X2, *a, __orig=orig, __name=name, **k):
            if __name != "colstats":
                seen[__name] += 1
            elif X2.shape[1] == P:
                seen["colstats_x"] += 1
            return __orig(self, X2, *a, **k)
        monkeypatch.setattr(HipBackend, name, counted)
    X = torch.from_numpy(x).cuda().to(torch.float32 if dtype == "float32" else torch.float64)
    keep = X.clone()
    one = api.tPLS(4, dtype=dtype, algorithm="xcov", copy_X=False)
    one.fit(X, y)
    assert seen == {"colstats_x": 0, "xcov_stats": 1, "xcov_ssq": 0}, seen
    assert one.fit_report_["stats_with_s"] is True and one.fit_report_["raw"] and torch.equal(X, keep)
    two = api.tPLS(4, dtype=dtype, algorithm="xcov", copy_X=False, options=default_options().but(xcov_stats_with_s=False))
    two.fit(X, y)
    assert seen == {"colstats_x": 1, "xcov_stats": 1, "xcov_ssq": 1} and two.fit_report_["stats_with_s"] is False
    assert one.n_iter_ == two.n_iter_
    for f, g in zip(one.X_factors + one.Y_factors, two.X_factors + two.Y_factors):
        assert _normwise(f, g) <= 1e-10
    assert_allclose(one.R2X, two.R2X, rtol=0, atol=1e-11)
    assert_allclose(one.X_mean, two.X_mean, rtol=0, atol=1e-12 * np.abs(two.X_mean).max())
    fit = O.fit_tpls(x, y, 4) if shape[0] > 200 or len(shape) == 2 else None
    if fit is not None:
        assert one.n_iter_ == fit.n_iter
        assert_allclose(one.R2X, fit.r2x[0], rtol=0, atol=1e-6 if dtype == "float32" else 1e-9)


def test_one_missing_value_the_probe_did_not_see_sends_the_fit_to_the_masked_statistics(api):
    """The optimistic read finds a NaN in a column sum (the <= 256-row probe had not met it): the regular statistics pass and the
    masked forms take over, the report says why, the fit equals the one that never tried."""
    x, y, _ = O.import_synthetic((600, 16, 24), 4, 3, error=0.1, seed=13)
    x[301, 5, 7] = np.nan                                          # (the probe takes rows 0, 2, 4, ...)
    m = api.tPLS(3, algorithm="xcov")
    m.fit(x, y)
    assert m.X_hasMiss and m.fit_report_["stats_with_s"] is False
    assert any("non-finite" in d for d in m.fit_report_["declined"]), m.fit_report_["declined"]
    ref = api.tPLS(3, algorithm="xcov", options=default_options().but(xcov_stats_with_s=False))
    ref.fit(x, y)
    assert m.n_iter_ == ref.n_iter_
    assert np.array_equal(m.X_factors[0], ref.X_factors[0])
