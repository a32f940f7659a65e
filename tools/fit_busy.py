#!/usr/bin/env python3
"""How much of a fit's wall time is the GPU busy?  Runs N whole fits (default: xcov, cfg-2 shape) and prints wall time per fit; run it
under `rocprofv3 --kernel-trace --stats` and divide the summed kernel time by N for the busy time per fit.
Usage: python tools/fit_busy.py [direct|xcov] [N] [two|one] [f64|nan]      (nan: 30 % missing values, BASELINE configs[3]; two: xcov with two reads of X per component, xcov_one_read = False)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmtf_pls_amd.backend import HipBackend
from cmtf_pls_amd.engine import EngineOptions, NipalsEngine
from cmtf_pls_amd.synthetic import synthetic_shard_device

algo = sys.argv[1] if len(sys.argv) > 1 else "xcov"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 5
opts = EngineOptions(xcov_one_read=not (len(sys.argv) > 3 and sys.argv[3] == "two"),
                     xcov_pipeline=not os.environ.get("CMTFPLS_NO_PIPELINE"))      # A/B of the pipelined inner loop
dev = torch.device("cuda:0")
eng = NipalsEngine(HipBackend(dev), None, opts)
f64 = len(sys.argv) > 4 and sys.argv[4] == "f64"          # f64 storage: half the rows, the same bytes
nan = len(sys.argv) > 4 and sys.argv[4] == "nan"
X, Y = synthetic_shard_device((32768 if f64 else 65536, 128, 128), 16, 10, error=0.1, seed=215, device=dev, dtype=torch.float64 if f64 else None,
                              nan_fraction=0.3 if nan else 0.0)
walls = []
for i in range(N + 1):
    Xf, Yf = X.clone(), Y.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st = eng.fit([Xf], Yf, 10, tol=1e-8, max_iter=100, coupled=False, algorithm=algo)
    torch.cuda.synchronize()
    walls.append(time.perf_counter() - t0)
print("last fit:", st.report, flush=True)
print(f"{algo}: first {walls[0]*1e3:.2f} ms, then {[round(w*1e3, 2) for w in walls[1:]]} ms per fit, {sum(st.n_iter)} iterations, {N + 1} fits in all", flush=True)
