#!/usr/bin/env python3
"""Randomised differential checks of what round 4 added (run on the GPU box; exits non-zero on the first mismatch):
  xcov_tiles   cmtfpls_xcov_* / xcov_ssq with 1..200 responses (tiles of 64) against torch's f64 matmul, plain and masked
  xcov_fit     algorithm="xcov" with M in 65..140 against the direct loop (tPLS / ctPLS, f32 / f64, with and without NaNs)
  project      transform of batches with a random fraction of incomplete samples, 1..4 coupled blocks of mixed orders and storage
               types, against the oracle's masked sequence; complete samples bit-equal to transforming them alone
  loo          cmtfpls_loo_xcov_f64 against cmtfpls_loo_tpls_f64 where both apply, and against literal refits of sampled folds
               beyond the LDS form (odd sizes: n not a multiple of 16, B < A, matrices, M not a multiple of 16)
  report       every fit's report is consistent with what was asked (algorithm, missing, responses)
Usage: python tools/fuzz_round4.py [cases-per-kind] [seed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as O  # noqa: E402
from cmtf_pls_amd import EngineOptions, ctPLS, tPLS  # noqa: E402
from cmtf_pls_amd.backend import HipBackend  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
be = HipBackend(torch.device("cuda:0"))
REG = EngineOptions(small_fit=False)                   # the kernels under test are the multi-launch engine's


def f32r(a):
    return a.astype(np.float32).astype(np.float64)


def normwise(got, want):
    scale = np.nanmax(np.abs(want), axis=0, keepdims=True)
    scale[scale == 0] = 1.0
    return float(np.nanmax(np.abs(got - want) / (np.abs(want) + scale)))


# ---- xcov response tiles -------------------------------------------------------------------------------------------------------
g = torch.Generator(device="cpu").manual_seed(int(rng.integers(1 << 30)))
for case in range(N):
    dt = [torch.float32, torch.float64][int(rng.integers(2))]
    I, P, M = int(rng.integers(1, 900)), int(rng.choice([4, 12, 64, 100, 256, 768, 1024, 4100])), int(rng.integers(1, 201))
    X = torch.randn(I, P, generator=g, dtype=torch.float64).to(dt).cuda()
    Y = torch.randn(I, M, generator=g, dtype=torch.float64).cuda()
    want = Y.T @ X.double()
    S = be.xcov(X, Y, False)
    err = float((S - want).abs().max() / max(float(want.abs().max()), 1e-300))
    assert err < 1e-11, ("xcov", I, P, M, dt, err)
    if P % 4 == 0:
        mean = X.double().mean(dim=0)
        S2, ssq = be.xcov_ssq(X, Y, mean, out=be.empty(M, P))
        assert torch.equal(S2, S), ("xcov_ssq S", I, P, M, dt)
        ref = float(((X.double() - mean) ** 2).sum())
        assert abs(float(ssq.item()) - ref) <= 1e-9 * max(ref, 1e-300), ("xcov_ssq norm", I, P, M, dt)
    Xn = X.clone()
    Xn[torch.rand(I, P, generator=g).cuda() < 0.2] = float("nan")
    Sm = be.xcov(Xn, Y, True)
    wantm = Y.T @ torch.nan_to_num(Xn.double(), nan=0.0)
    err = float((Sm - wantm).abs().max() / max(float(wantm.abs().max()), 1e-300))
    assert err < 1e-11, ("xcov masked", I, P, M, dt, err)
print(f"xcov response tiles: {N} random cases ok", flush=True)

# ---- xcov fits with more than 64 responses -----------------------------------------------------------------------------------------
n_fit = max(4, N // 4)
for case in range(n_fit):
    f32 = bool(rng.integers(2))
    coupled = bool(rng.integers(2))
    nan = bool(rng.integers(2))
    I, A, B = int(rng.integers(40, 200)), int(rng.choice([4, 8, 16, 32])), int(rng.choice([16, 32, 64, 128]))
    M, R = int(rng.integers(65, 141)), int(rng.integers(1, 4))
    lat = rng.normal(size=(I, 4))
    x = np.einsum("ir,jr,kr->ijk", lat, rng.normal(size=(A, 4)), rng.normal(size=(B, 4))) + 0.2 * rng.normal(size=(I, A, B))
    y = lat @ rng.normal(size=(4, M)) + 0.2 * rng.normal(size=(I, M))
    blocks = [x] + ([lat @ rng.normal(size=(4, 24)) + 0.2 * rng.normal(size=(I, 24))] if coupled else [])
    if f32:
        blocks, y = [f32r(b) for b in blocks], f32r(y)
    if nan:
        blocks[0][rng.random(blocks[0].shape) < 0.15] = np.nan
    dtype = "float32" if f32 else "float64"
    make = lambda alg: (ctPLS if coupled else tPLS)(R, dtype=dtype, algorithm=alg, options=REG)
    a, d = make("xcov"), make("direct")
    for m in (a, d):
        m.fit(blocks if coupled else blocks[0], y, max_iter=30)
    rep = a.fit_report_
    assert rep["algorithm"] == "xcov" and rep["responses"] == M and rep["missing"][0] == nan, rep
    assert f"{(M + 63) // 64} response tiles" in rep["s_build"], rep
    Ta, Td = (a.factor_T, d.factor_T) if coupled else (a.X_factors[0], d.X_factors[0])
    assert max(abs(p - q) for p, q in zip(a.n_iter_, d.n_iter_)) <= (1 if f32 else 0), ("xcov M>64 n_iter", a.n_iter_, d.n_iter_)
    if list(a.n_iter_) == list(d.n_iter_):
        e = normwise(Ta, Td)
        assert e < (5e-5 if f32 else 1e-7), ("xcov M>64 vs direct", (I, A, B), M, R, dtype, coupled, nan, e)
print(f"xcov fits with 65..140 responses: {n_fit} random fits ok", flush=True)


# ---- projection with a random fraction of incomplete samples -------------------------------------------------------------------------
def oracle_fit_of(m, coupled):
    if coupled:
        loads, means, shapes, T = [list(f[1:]) for f in m.Xs_factors], list(m.Xs_mean), list(m.Xs_shape), m.factor_T
    else:
        loads, means, shapes, T = [list(m.X_factors[1:])], [m.X_mean], [m.X_shape], m.X_factors[0]
    R = m.n_components
    return O.OracleFit(coupled=coupled, n_components=R, block_shapes=shapes, y_shape=m.Y_shape, T=T, loadings=loads, U=m.Y_factors[0],
                       Q=m.Y_factors[1], coef=m.coef_, r2x=[np.zeros(R)] * len(loads), r2y=m.R2Y, x_means=means, y_mean=m.Y_mean,
                       has_miss=[False] * len(loads))


forms = {}
n_proj = max(6, N // 2)
for case in range(n_proj):
    nb = int(rng.integers(1, 5))
    f32 = bool(rng.integers(2))
    I, R, M = int(rng.integers(20, 60)), int(rng.integers(1, 6)), int(rng.integers(1, 5))
    lat = rng.normal(size=(I, 3))
    blocks = []
    for b in range(nb):
        kind = int(rng.integers(3)) if b else 1
        if kind == 0:
            sh = (int(rng.choice([16, 64, 512, 1000])),)
        elif kind == 1:
            sh = (int(rng.choice([2, 4, 8, 16, 32, 6])), int(rng.choice([8, 16, 64, 128, 256])))
        else:
            sh = (int(rng.choice([2, 4])), int(rng.choice([4, 8])), int(rng.choice([8, 16])))
        fac = [rng.normal(size=(d, 3)) for d in sh]
        sub = "".join("jkl"[i] + "r," for i in range(len(sh)))[:-1]
        dense = np.einsum("ir," + sub + "->i" + "jkl"[:len(sh)], lat, *fac) + 0.2 * rng.normal(size=(I,) + sh)
        blocks.append(f32r(dense) if f32 else dense)
    y = lat @ rng.normal(size=(3, M)) + 0.2 * rng.normal(size=(I, M))
    coupled = nb > 1
    dtype = "float32" if f32 else "float64"
    m = (ctPLS if coupled else tPLS)(R, dtype=dtype, options=REG)
    m.fit(blocks if coupled else blocks[0], y, max_iter=15)
    n_new = min(I, int(rng.integers(1, 40)))
    new = [b[:n_new].copy() for b in blocks]
    frac = float(rng.choice([0.0, 0.1, 0.5, 1.0]))
    bad = rng.random(n_new) < frac
    for b in new:
        hole = rng.random(b.shape) < 0.3
        hole[~bad] = False
        b[hole] = np.nan
    if bad.any() and rng.random() < 0.5:
        new[int(rng.integers(nb))][np.flatnonzero(bad)[0]] = np.nan                 # an empty row in one block
    arg = new if coupled else new[0]
    got = m.transform(arg)
    form = m.projection_report_["form"]
    forms[form] = forms.get(form, 0) + 1
    want = O.transform(oracle_fit_of(m, coupled), arg)
    assert np.array_equal(np.isnan(got), np.isnan(want)), ("nan pattern", [b.shape for b in blocks], dtype, frac, form)
    ok = ~np.isnan(want).any(axis=1)
    if ok.any():
        err = normwise(got[ok], want[ok])
        assert err < (5e-5 if f32 else 1e-8), ("project", [b.shape for b in blocks], dtype, R, frac, form, err)
    anynan = np.zeros(n_new, dtype=bool)
    for b in new:
        anynan |= np.isnan(b.reshape(n_new, -1)).any(axis=1)
    if (~anynan).any() and anynan.any() and "one-pass MTTKRP for the complete samples" in form:
        alone = m.transform([b[~anynan] for b in new] if coupled else new[0][~anynan])
        assert np.array_equal(got[~anynan], alone), ("complete samples changed by the batch", [b.shape for b in blocks], dtype, form)
print(f"projection of partly incomplete batches: {n_proj} random cases ok; forms taken: {forms}", flush=True)

# ---- leave-one-out: xcov kernel vs LDS kernel, and vs literal refits -----------------------------------------------------------------
n_loo = max(6, N // 3)
both = 0
for case in range(n_loo):
    order3 = rng.random() < 0.8
    I = int(rng.integers(12, 60))
    A, B = (int(rng.integers(2, 90)), int(rng.integers(2, 90))) if order3 else (1, int(rng.integers(2, 400)))
    M, R = int(rng.integers(1, 20)), int(rng.integers(1, 5))
    R = min(R, I - 2)
    shape = (I, A, B) if order3 else (I, B)
    x, y, _ = O.import_synthetic(shape, M, 3, error=0.3, seed=int(rng.integers(1 << 30)))
    X2 = torch.from_numpy(x.reshape(I, -1)).cuda()
    Y = torch.from_numpy(y.reshape(I, -1)).cuda()
    xc = be.loo_tpls(X2, Y, A, B, R, 1e-8, 100, forms=("xcov",))
    assert xc is not None, ("loo_xcov declined", shape, M, R)
    lds = be.loo_tpls(X2, Y, A, B, R, 1e-8, 100, forms=("lds",))
    if lds is not None:
        both += 1
        same_iters = torch.equal(lds[1], xc[1])
        err = float((lds[0] - xc[0]).abs().max() / max(float(lds[0].abs().max()), 1e-300))
        assert err < (1e-8 if same_iters else 1e-6), ("loo xcov vs lds", shape, M, R, err, same_iters)
    else:
        for i in (0, I - 1):
            keep = np.ones(I, dtype=bool)
            keep[i] = False
            r = tPLS(R, options=REG)
            r.fit(x[keep], y[keep])
            want = r.predict(x[i:i + 1]).reshape(-1)
            got = xc[0][i].cpu().numpy().reshape(-1)
            err = float(np.abs(got - want).max() / max(1.0, np.abs(want).max()))
            assert err < 1e-7, ("loo xcov vs refit", shape, M, R, i, err)
print(f"leave-one-out: {n_loo} random shapes ok ({both} against the LDS kernel, the rest against literal refits)", flush=True)
print("fuzz_round4: all ok", flush=True)
