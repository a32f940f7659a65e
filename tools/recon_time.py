#!/usr/bin/env python3
"""X_reconstructed / R2X_literal at BASELINE configs[1] (65536 x 128 x 128 f32): device tensor, host float64 array, literal R2X."""
import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmtf_pls_amd import tPLS
from cmtf_pls_amd.synthetic import synthetic_shard_device
X, Y = synthetic_shard_device((65536, 128, 128), 16, 10, error=0.1, device="cuda:0")
m = tPLS(3, dtype="float32"); m.fit(X, Y, max_iter=10)
m.X_reconstructed(rows=slice(0, 64))
torch.cuda.synchronize(); t0 = time.perf_counter(); d = m.X_reconstructed(device=True); torch.cuda.synchronize(); print("X_reconstructed(device=True) 4.3 GB:", round(time.perf_counter() - t0, 4), "s")
t0 = time.perf_counter(); h = m.X_reconstructed(); print("X_reconstructed() -> host float64 8.6 GB:", round(time.perf_counter() - t0, 3), "s", h.dtype, h.shape)
print("R2X literal on device:", m.R2X_literal(X), "booked:", m.R2X[-1])
