#!/usr/bin/env python3
"""Fit / transform at BASELINE configs[1] with the input on the HOST (NumPy float32, pageable): what a
caller of the drop-in API pays including the PCIe copy.  Never the bench's `value`."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmtf_pls_amd import tPLS
from cmtf_pls_amd.synthetic import synthetic_shard_device

X, Y = synthetic_shard_device((65536, 128, 128), 16, 10, error=0.1, device="cuda:0")
xh, yh = X.cpu().numpy(), Y.cpu().numpy()
for algo in ("direct", "xcov"):
    m = tPLS(10, dtype="float32", algorithm=algo)
    m.fit(X, Y)                                    # warm-up: first launches load code objects, size workspaces
    torch.cuda.synchronize(); t0 = time.perf_counter(); m.fit(xh, yh); torch.cuda.synchronize()
    th = time.perf_counter() - t0
    torch.cuda.synchronize(); t0 = time.perf_counter(); m.fit(X, Y); torch.cuda.synchronize()
    td = time.perf_counter() - t0
    print(f"fit R=10 {algo:6s}: host NumPy input {th:.3f} s   device input {td:.3f} s   (X = {xh.nbytes/1e9:.2f} GB)")
m.transform(X)                                     # warm-up
torch.cuda.synchronize(); t0 = time.perf_counter(); T = m.transform(X); torch.cuda.synchronize()
print(f"transform (one MTTKRP pass) device input: {time.perf_counter()-t0:.3f} s")
torch.cuda.synchronize(); t0 = time.perf_counter(); T = m.transform(xh); torch.cuda.synchronize()
print(f"transform host input: {time.perf_counter()-t0:.3f} s")
