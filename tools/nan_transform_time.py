#!/usr/bin/env python3
"""transform / predict of samples WITH missing values: the row-in-registers form (cmtfpls_project_rows_* /
cmtfpls_project_rows2_*: ONE read of the raw blocks, nothing written) against the passes it replaces (clone + centre +
R x score / average / deflate), at
  BASELINE configs[3]        65536 x 128 x 128 f32, 30 % NaN           (256-thread workgroups, 16 vectors per lane)
  BASELINE configs[4]-shaped 32768 x 256 x 256 f32, 30 % NaN           (1024-thread workgroups, round 3)
  BASELINE configs[2]        65536 x 128 x 128 + 65536 x 512, coupled  (both rows of a sample in one workgroup, round 3)
  rows30                     65536 x 128 x 128 f32 fitted on complete data; the batch has 30 % of its SAMPLES incomplete (each with
                             30 % of its entries missing): complete samples keep their one-pass MTTKRP scores, the incomplete ones take
                             the masked sequence (round 4, EngineOptions.project_split_rows) -- against every row through the sequence
  coupled3                   65536 x 128 x 128 + 65536 x 512 + 65536 x 16 x 8, 10 % of the samples incomplete: three coupled blocks
                             (round 4: MTTKRP + the sequential passes on compact copies of the incomplete samples)
Usage: python tools/nan_transform_time.py [cfg3|cfg5|coupled|rows30|coupled3 ...]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmtf_pls_amd import ctPLS, tPLS
from cmtf_pls_amd.synthetic import synthetic_shard_device
from cmtf_pls_amd.tpls import to_device_copy


def clock(fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, out


def report(name, m, Xs, gbytes):
    eng = m._get_engine()
    arg = Xs if isinstance(m, ctPLS) else Xs[0]
    t_new, s_new = clock(lambda: m.transform(arg))
    t_ro, s_ro = clock(lambda: eng.project_readonly(m._state, Xs))
    assert s_ro is not None, "the one-read form declined this shape"
    t_old, s_old = clock(lambda: eng.project(m._state, [to_device_copy(X, torch.float32, "cuda:0") for X in Xs]).cpu().numpy())
    d = np.nanmax(np.abs(s_new - s_old)) / np.nanmax(np.abs(s_old))
    print(f"{name}: transform API {t_new*1e3:.2f} ms (engine.project_readonly {t_ro*1e3:.2f} ms = MTTKRP attempt + NaN flag + "
          f"row-in-registers kernel, {gbytes / t_ro / 1e3:.2f} TB/s of X) | clone + centre + 10 passes {t_old*1e3:.2f} ms | "
          f"max |diff| / max|scores| = {d:.1e}", flush=True)
    t_p, _ = clock(lambda: m.predict(arg))
    print(f"{name}: predict {t_p*1e3:.2f} ms", flush=True)


def split_report(name, m, Xs, gbytes):
    from cmtf_pls_amd.engine import EngineOptions, NipalsEngine
    eng = m._get_engine()
    every = NipalsEngine(eng.be, None, EngineOptions(project_split_rows=False))
    t_split, s_split = clock(lambda: eng.project_readonly(m._state, Xs))
    rep = dict(eng.last_projection)
    t_every, s_every = clock(lambda: every.project_readonly(m._state, Xs))
    t_old, s_old = clock(lambda: eng.project(m._state, [to_device_copy(X, torch.float32, "cuda:0") for X in Xs]))
    ok = ~torch.isnan(s_old).any(dim=1)
    d = float((s_split[ok] - s_old[ok]).abs().max() / s_old[ok].abs().max())
    print(f"{name}: per-sample form {t_split*1e3:.2f} ms ({gbytes / t_split / 1e3:.2f} TB/s of X; {rep.get('incomplete_rows')} of {rep.get('rows')} samples "
          f"incomplete; {rep.get('form')}) | every row through the masked sequence {'%.2f ms' % (t_every*1e3) if s_every is not None else 'n/a'} | "
          f"clone + centre + 10 passes {t_old*1e3:.2f} ms | max |diff| / max|scores| = {d:.1e}", flush=True)


which = sys.argv[1:] or ["cfg3", "cfg5", "coupled", "rows30", "coupled3"]
if "rows30" in which or "coupled3" in which:
    from cmtf_pls_amd.backend import HipBackend
    be0 = HipBackend(torch.device("cuda:0"))
    X, Y, Xm = synthetic_shard_device((65536, 128, 128), 16, 10, error=0.1, device="cuda:0", matrix_block=512, seed=217)

    def punch(T, frac_rows, seed):
        """30 % missing entries in a random `frac_rows` of the samples."""
        holes = T.clone()
        be0.add_noise(holes.view(holes.shape[0], -1), 0.0, seed, 0, 0.3)
        keep = torch.rand(T.shape[0], device=T.device, generator=torch.Generator(device=T.device).manual_seed(seed)) >= frac_rows
        holes[keep] = T[keep]
        return holes
    if "rows30" in which:
        m = tPLS(10, dtype="float32", algorithm="xcov")
        m.fit(X, Y, max_iter=30)
        split_report("65536 x 128 x 128 f32, 30 % of the SAMPLES incomplete, R = 10", m, [punch(X, 0.3, 5)], X.numel() * 4 / 1e9)
        split_report("65536 x 128 x 128 f32, 3 % of the samples incomplete, R = 10", m, [punch(X, 0.03, 6)], X.numel() * 4 / 1e9)
    if "coupled3" in which:
        X3 = torch.randn(65536, 16, 8, device="cuda:0") + Y[:, :1].float()[:, :, None]
        m = ctPLS(10, dtype="float32", algorithm="xcov")
        m.fit([X, Xm, X3], Y, max_iter=30)
        gb = (X.numel() + Xm.numel() + X3.numel()) * 4 / 1e9
        split_report("three coupled blocks (128x128, 512, 16x8), 10 % of the samples incomplete, R = 10", m,
                     [punch(X, 0.1, 7), punch(Xm, 0.1, 8), X3], gb)
    del X, Y, Xm
if "cfg3" in which:
    X, Y = synthetic_shard_device((65536, 128, 128), 16, 10, error=0.1, device="cuda:0", nan_fraction=0.3, seed=217)
    m = tPLS(10, dtype="float32", algorithm="xcov")
    m.fit(X, Y, max_iter=30)
    report("65536 x 128 x 128 f32, 30 % NaN, R = 10", m, [X], X.numel() * 4 / 1e9)
    del X, Y, m
if "cfg5" in which:
    X, Y = synthetic_shard_device((32768, 256, 256), 32, 10, error=0.1, device="cuda:0", nan_fraction=0.3, seed=217)
    m = tPLS(10, dtype="float32", algorithm="xcov")
    m.fit(X, Y, max_iter=30)
    report("32768 x 256 x 256 f32, 30 % NaN, R = 10", m, [X], X.numel() * 4 / 1e9)
    del X, Y, m
if "coupled" in which:
    X, Y, Xm = synthetic_shard_device((65536, 128, 128), 16, 10, error=0.1, device="cuda:0", matrix_block=512, nan_fraction=0.3, seed=217)
    from cmtf_pls_amd.backend import HipBackend
    HipBackend(torch.device("cuda:0")).add_noise(Xm, 0.0, 991, 0, 0.3)          # the matrix block gets its own 30 % NaN mask
    m = ctPLS(10, dtype="float32", algorithm="xcov")
    m.fit([X, Xm], Y, max_iter=30)
    report("65536 x 128 x 128 + 65536 x 512 f32 coupled, 30 % NaN, R = 10", m, [X, Xm], (X.numel() + Xm.numel()) * 4 / 1e9)
