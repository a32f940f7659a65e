#!/usr/bin/env python3
"""Fit / transform at BASELINE configs[1] with the input on the HOST (NumPy, pageable): what a caller of the drop-in
API pays including the PCIe copy -- float32 input, and float64 input (the reference's natural type) stored as float32.
Never the bench's `value`."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmtf_pls_amd import tPLS
from cmtf_pls_amd.synthetic import synthetic_shard_device


def clock(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize()
    return time.perf_counter() - t0, out


X, Y = synthetic_shard_device((65536, 128, 128), 16, 10, error=0.1, device="cuda:0")
xh, yh = X.cpu().numpy(), Y.cpu().numpy()
xh64 = xh.astype(np.float64)
for algo in ("direct", "xcov"):
    m = tPLS(10, dtype="float32", algorithm=algo)
    m.fit(X, Y)                                    # warm-up: first launches load code objects, size workspaces
    th, _ = clock(lambda: m.fit(xh, yh))
    th64, _ = clock(lambda: m.fit(xh64, yh))
    td, _ = clock(lambda: m.fit(X, Y))
    print(f"fit R=10 {algo:6s}: host f32 input {th:.3f} s   host f64 input (f32 storage) {th64:.3f} s   device input {td:.3f} s   "
          f"(X = {xh.nbytes/1e9:.2f} GB f32 / {xh64.nbytes/1e9:.2f} GB f64)", flush=True)
m.transform(X)                                     # warm-up
t, _ = clock(lambda: m.transform(X)); print(f"transform (one MTTKRP pass) device input: {t:.4f} s")
t, _ = clock(lambda: m.transform(xh)); print(f"transform host f32 input: {t:.4f} s")
t, _ = clock(lambda: m.transform(xh64)); print(f"transform host f64 input: {t:.4f} s")
t, _ = clock(lambda: m.predict(X)); print(f"predict device input: {t:.4f} s")
m.transform(X, yh)
t, _ = clock(lambda: m.transform(X, yh)); print(f"transform(X, Y) device X, host Y (Y side through rowdot / y_deflate): {t:.4f} s")
# where transform's time goes (device input): the read-only one-pass form against the copy + centre + MTTKRP it replaced
from cmtf_pls_amd.tpls import to_device_copy
eng = m._get_engine()
t, sc = clock(lambda: eng.project_readonly(m._state, [X])); print(f"  engine.project_readonly (MTTKRP on the caller's X + shift + fix-up): {t:.4f} s")
t, Xd = clock(lambda: to_device_copy(X, torch.float32, eng.be.device)); print(f"  [old path] device clone of X: {t:.4f} s")
t, sc2 = clock(lambda: eng.project(m._state, [Xd])); print(f"  [old path] engine.project (centre + MTTKRP + fix-up): {t:.4f} s")
print(f"  max |readonly - old| / max|scores| = {float((sc - sc2).abs().max() / sc2.abs().max()):.2e}")
t, _ = clock(lambda: sc.cpu().numpy()); print(f"  scores to host: {t:.4f} s")
