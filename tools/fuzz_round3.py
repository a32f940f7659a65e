#!/usr/bin/env python3
"""Randomised differential checks of the kernels added in round 3 (run on the GPU box; exits non-zero on the first mismatch):
  mttkrp       every form (k-row on 4x4x4 / 16x16x4 MFMAs, j-block, tile) at random shapes against NumPy
  project      one-read NaN transform (256- and 1024-thread forms, two coupled blocks) against the oracle's masked sequence
  fit_small    the one-launch fit against the regular engine
Usage: python tools/fuzz_round3.py [cases-per-kind] [seed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as O  # noqa: E402
from cmtf_pls_amd import ctPLS, tPLS  # noqa: E402
from cmtf_pls_amd.backend import HipBackend  # noqa: E402
from cmtf_pls_amd.engine import EngineOptions, set_default_options  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
be = HipBackend(torch.device("cuda:0"))
dev = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0").to(dt)


def normwise(got, want):
    scale = np.nanmax(np.abs(want), axis=0, keepdims=True)
    scale[scale == 0] = 1.0
    return float(np.nanmax(np.abs(got - want) / (np.abs(want) + scale)))


# ---- mttkrp ----------------------------------------------------------------------------------------------------------------
for case in range(N):
    dt = [torch.float32, torch.float64][int(rng.integers(2))]
    A = int(rng.choice([4, 8, 12, 16, 20, 32, 48, 64, 96, 128, 10, 1]))
    B = int(rng.choice([64, 128, 192, 256, 320, 384, 48, 32, 16, 100]))
    I = int(rng.integers(1, 70))
    R = int(rng.integers(1, 18))
    x = rng.normal(size=(I, A * B))
    if dt == torch.float32:
        x = x.astype(np.float32).astype(np.float64)
    WA, WB = rng.normal(size=(A, R)), rng.normal(size=(B, R))
    out = be.mttkrp(dev(x, dt), A, B, dev(WA), dev(WB), be.empty(I, R))
    want = x @ (WA[:, None, :] * WB[None, :, :]).reshape(A * B, R)
    if out is None:
        assert (A + B) * 16 * ((R + 15) // 16) * 8 > 152 * 1024, ("mttkrp declined", I, A, B, R)
        continue
    err = np.abs(out.cpu().numpy() - want).max() / max(np.abs(want).max(), 1e-300)
    assert err < 1e-11, ("mttkrp", I, A, B, R, dt, err)
print(f"mttkrp: {N} random cases ok", flush=True)

# ---- one-read NaN projection -----------------------------------------------------------------------------------------------
def oracle_fit_of(m, coupled):
    if coupled:
        loads, means, shapes, T = [list(f[1:]) for f in m.Xs_factors], list(m.Xs_mean), list(m.Xs_shape), m.factor_T
    else:
        loads, means, shapes, T = [list(m.X_factors[1:])], [m.X_mean], [m.X_shape], m.X_factors[0]
    R = m.n_components
    return O.OracleFit(coupled=coupled, n_components=R, block_shapes=shapes, y_shape=m.Y_shape, T=T, loadings=loads, U=m.Y_factors[0],
                       Q=m.Y_factors[1], coef=m.coef_, r2x=[np.zeros(R)] * len(loads), r2y=m.R2Y, x_means=means, y_mean=m.Y_mean,
                       has_miss=[False] * len(loads))


calls = {"rows": 0, "rows2": 0}
for name, key in (("project_rows", "rows"), ("project_rows2", "rows2")):
    orig = getattr(HipBackend, name)

    def wrapped(self, *a, __orig=orig, __key=key, **k):
        out = __orig(self, *a, **k)
        calls[__key] += out is not None
        return out
    setattr(HipBackend, name, wrapped)

set_default_options(EngineOptions(small_fit=False))
for case in range(N // 2):
    f32 = bool(rng.integers(2))
    coupled = bool(rng.integers(2))
    A = int(rng.choice([1, 2, 4, 8, 16, 32, 64, 128, 256, 6]))
    B = int(rng.choice([8, 16, 32, 64, 128, 256, 512, 1024]))
    if A * B > (65536 if f32 else 32768):
        B = 64
    I, R, M = int(rng.integers(12, 40)), int(rng.integers(1, 7)), int(rng.integers(1, 5))
    shape = (I, B) if A == 1 else (I, A, B)
    x, y, cp = O.import_synthetic(shape, M, 3, error=0.2, seed=int(rng.integers(1 << 30)))
    blocks = [x]
    if coupled:
        Bm = int(rng.choice([16, 64, 512, 1024]))
        blocks.append(cp.factors[0] @ rng.normal(size=(Bm, 3)).T + 0.2 * rng.normal(size=(I, Bm)))
    if f32:
        blocks = [b.astype(np.float32).astype(np.float64) for b in blocks]
        y = y.astype(np.float32).astype(np.float64)
    dtype = "float32" if f32 else "float64"
    m = ctPLS(R, dtype=dtype) if coupled else tPLS(R, dtype=dtype)
    m.fit(blocks if coupled else blocks[0], y, max_iter=15)
    new = [b[:10].copy() for b in blocks]
    for b in new:
        b[rng.random(b.shape) < 0.25] = np.nan
    new[0][3] = np.nan                                                   # an empty row
    got = m.transform(new if coupled else new[0])
    want = O.transform(oracle_fit_of(m, coupled), new if coupled else new[0])
    assert np.array_equal(np.isnan(got), np.isnan(want)), ("nan pattern", shape, coupled, dtype)
    ok = ~np.isnan(want).any(axis=1)
    err = normwise(got[ok], want[ok])
    assert err < (2e-5 if f32 else 1e-8), ("project", shape, coupled, dtype, R, err)
print(f"one-read NaN projection: {N // 2} random cases ok ({calls['rows']} through project_rows, {calls['rows2']} through project_rows2, "
      f"the rest through the passes)", flush=True)

# ---- one-launch small fit ----------------------------------------------------------------------------------------------------
n_small = 0
for case in range(N // 3):
    order3 = bool(rng.integers(2))
    I = int(rng.integers(8, 120))
    A, B = (int(rng.integers(2, 20)), int(rng.integers(2, 20))) if order3 else (1, int(rng.integers(2, 200)))
    M, R = int(rng.integers(1, 6)), int(rng.integers(1, 7))
    R = min(R, A * B, I - 1)               # (beyond the rank of the centred X the loop iterates on rounding noise: nothing to compare)
    shape = (I, A, B) if order3 else (I, B)
    x, y, _ = O.import_synthetic(shape, M, 3, error=0.2, seed=int(rng.integers(1 << 30)))
    one = tPLS(R, options=EngineOptions(small_fit=True))
    one.fit(x, y)
    reg = tPLS(R, options=EngineOptions(small_fit=False))
    reg.fit(x, y)
    n_small += 1
    assert one.n_iter_ == reg.n_iter_, ("fit_small n_iter", shape, M, R, one.n_iter_, reg.n_iter_)
    for f, g in zip(one.X_factors + one.Y_factors, reg.X_factors + reg.Y_factors):
        e = normwise(f, g)
        assert e < 1e-7, ("fit_small", shape, M, R, e)
    assert np.abs(np.asarray(one.R2X) - reg.R2X).max() < 1e-10 and np.abs(np.asarray(one.R2Y) - reg.R2Y).max() < 1e-10
print(f"one-launch small fit: {n_small} random cases ok", flush=True)

# ---- score + contraction in one read (scorecontract.hip) -----------------------------------------------------------------------
n_sc = 0
for case in range(N):
    dt = [torch.float32, torch.float64][int(rng.integers(2))]
    V = 4 if dt == torch.float32 else 2
    A = int(rng.choice([1, 2, 3, 8, 16, 50, 64, 128]))
    B = V * int(rng.integers(1, 2048 // V + 1))
    I = int(rng.integers(1, 600))
    x = rng.normal(size=(I, A * B)) + 1.5
    if dt == torch.float32:
        x = x.astype(np.float32).astype(np.float64)
    wA, wB = rng.normal(size=A), rng.normal(size=B)
    use = rng.integers(2, size=3)
    sh, sub, oth = (np.array([rng.normal()]) if use[0] else None), (rng.normal(size=I) if use[1] else None), (rng.normal(size=I) if use[2] else None)
    alpha = float(rng.choice([1.0, 0.5, 1.0 / 3.0]))
    t, Z = be.empty(I), be.empty(A * B)
    out = be.score_contract(dev(x, dt), A, B, dev(wA), dev(wB), None if sh is None else dev(sh), t, Z,
                            sub_own=None if sub is None else dev(sub), add_other=None if oth is None else dev(oth), alpha=alpha)
    P = A * B
    if out is None:
        lim = 16 * (16384 if (V == 4 and (1024 * V) % B == 0) else 8192)                       # (round 4: up to 16 workgroups share a row)
        assert P > lim or P < 512 * V, ("score_contract declined", I, A, B, dt)
        continue
    n_sc += 1
    want_t = x @ np.kron(wA, wB) - (0.0 if sh is None else sh[0]) - (0.0 if sub is None else sub)
    want_Z = x.T @ (alpha * (want_t + (0.0 if oth is None else oth)))
    assert np.abs(t.cpu().numpy() - want_t).max() <= 1e-11 * max(np.abs(want_t).max(), 1e-300), ("score_contract t", I, A, B, dt)
    assert np.abs(Z.cpu().numpy() - want_Z).max() <= 1e-11 * max(np.abs(want_Z).max(), 1e-300), ("score_contract Z", I, A, B, dt)
print(f"score_contract: {n_sc} random cases ok ({N - n_sc} declined by shape)", flush=True)

# ---- the cross-covariance loop (one read per component, pipelined inner loop, paired S build) against the direct loop -------------
n_x = 0
for case in range(N // 3):
    f32 = bool(rng.integers(2))
    kind = str(rng.choice(["tensor", "coupled", "nan", "coupled_nan", "matrix"]))
    I = int(rng.integers(40, 400))
    A, B = int(rng.choice([8, 16, 32, 64])), int(rng.choice([32, 64, 128, 256]))
    M, R = int(rng.integers(1, 9)), int(rng.integers(1, 6))
    x, y, cp = O.import_synthetic((I, A, B), M, max(R, 2), error=0.3, seed=int(rng.integers(1 << 30)))
    x = x + float(rng.choice([0.0, 5.0]))
    xm = cp.factors[0] @ rng.normal(size=(int(rng.choice([16, 96, 512])), max(R, 2))).T + 0.3 * rng.normal(size=(I, 1))
    if "nan" in kind:
        x[rng.random(x.shape) < 0.2] = np.nan
    blocks = {"tensor": [x], "nan": [x], "matrix": [xm], "coupled": [xm, x], "coupled_nan": [x, xm]}[kind]
    if f32:
        blocks, y = [b.astype(np.float32).astype(np.float64) for b in blocks], y.astype(np.float32).astype(np.float64)
    dtype = "float32" if f32 else "float64"
    coupled = len(blocks) > 1
    arg = blocks if coupled else blocks[0]
    d = (ctPLS if coupled else tPLS)(R, dtype=dtype)
    d.fit(arg, y)
    xc = (ctPLS if coupled else tPLS)(R, dtype=dtype, algorithm="xcov")
    xc.fit(arg, y)
    Td, Tx = (d.factor_T, xc.factor_T) if coupled else (d.X_factors[0], xc.X_factors[0])
    if d.n_iter_ != xc.n_iter_:      # an iteration count may differ by one when |du| lands within rounding of tol
        assert max(abs(a - b) for a, b in zip(d.n_iter_, xc.n_iter_)) <= 1, ("xcov n_iter", kind, (I, A, B), M, R, d.n_iter_, xc.n_iter_)
        continue
    n_x += 1
    e = normwise(Tx, Td)
    assert e < (5e-5 if f32 else 1e-7), ("xcov vs direct", kind, (I, A, B), M, R, dtype, e)
    assert np.abs(np.asarray(xc.R2Y) - d.R2Y).max() < (1e-5 if f32 else 1e-9), ("xcov R2Y", kind)
print(f"xcov (one read, pipelined, paired S build) vs direct: {n_x} random fits ok; last fit's pipeline {xc.fit_report_.get('pipeline')}", flush=True)
