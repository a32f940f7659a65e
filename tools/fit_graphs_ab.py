#!/usr/bin/env python3
"""sec-to-fit at a benchmark shape, eager vs HIP-graph replay of the per-iteration launch sequences,
for both algorithms.  Usage: python tools/fit_graphs_ab.py [I J K] [--M 16] [--R 10]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmtf_pls_amd.backend import HipBackend  # noqa: E402
from cmtf_pls_amd.engine import NipalsEngine  # noqa: E402
from cmtf_pls_amd.synthetic import synthetic_shard_device  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("shape", nargs="*", type=int, default=[65536, 128, 128])
    ap.add_argument("--M", type=int, default=16)
    ap.add_argument("--R", type=int, default=10)
    args = ap.parse_args()
    I, J, K = args.shape
    dev = torch.device("cuda:0")
    be = HipBackend(dev)
    eng = NipalsEngine(be, None)
    X, Y = synthetic_shard_device((I, J, K), args.M, args.R, error=0.1, seed=215, device=dev, dtype=torch.float32)
    for algo in ("direct", "xcov"):
        for graphs in (False, True, False, True):
            Xf, Yf = X.clone(), Y.clone()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            st = eng.fit([Xf], Yf, args.R, tol=1e-8, max_iter=100, coupled=False, algorithm=algo, use_graphs=graphs)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(f"{algo:7s} graphs={int(graphs)}  {dt*1e3:8.2f} ms  iters={sum(st.n_iter)}  R2Y={st.r2y[-1]:.6f}", flush=True)
            del Xf, Yf


if __name__ == "__main__":
    main()
