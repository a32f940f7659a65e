#!/usr/bin/env python3
"""Wall time of an xcov fit at cfg-2 by phase (begin / start_component / inner loop / finish_component / result), with a device
synchronisation at every phase boundary (so the sum is a little above the un-instrumented fit).
Usage: python tools/fit_phases.py [xcov|direct] [N]      CMTFPLS_NO_PIPELINE=1: the waiting inner loop"""
import os
import sys
import time
from collections import defaultdict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmtf_pls_amd import engine as E  # noqa: E402
from cmtf_pls_amd.backend import HipBackend  # noqa: E402
from cmtf_pls_amd.synthetic import synthetic_shard_device  # noqa: E402

algo = sys.argv[1] if len(sys.argv) > 1 else "xcov"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4
OPTS = E.EngineOptions(xcov_pipeline=not os.environ.get("CMTFPLS_NO_PIPELINE"))
acc = defaultdict(float)


def timed(cls, name):
    orig = getattr(cls, name)

    def wrapper(self, *a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = orig(self, *a, **k)
        torch.cuda.synchronize()
        acc[name] += time.perf_counter() - t0
        return out
    setattr(cls, name, wrapper)


for name in ("start_component", "inner_loop", "finish_component", "result"):
    timed(E.FitRun, name)
timed(E.NipalsEngine, "begin")
dev = torch.device("cuda:0")
eng = E.NipalsEngine(HipBackend(dev), None, OPTS)
X, Y = synthetic_shard_device((65536, 128, 128), 16, 10, error=0.1, seed=215, device=dev)
for i in range(N + 1):
    if i == 1:
        acc.clear()
    Yf = Y.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st = eng.fit([X], Yf, 10, tol=1e-8, max_iter=100, coupled=False, algorithm=algo, owned=[False])
    torch.cuda.synchronize()
    acc["fit"] += time.perf_counter() - t0
print(f"{algo}, {sum(st.n_iter)} iterations, ms per fit over {N} fits:", {k: round(1e3 * v / N, 3) for k, v in acc.items()}, flush=True)
