#!/usr/bin/env python3
"""Why the two read sweeps of the direct iteration run below their large-size rate on the 8192-row shard a GPU sees at N = 8
(VERDICT r3 "Next" #4): time(I) = t0 + bytes(I) / BW for the contraction (both kernels) and the score, I = 512 ... 65536 rows
of 128 x 128 f32.  Each point: a HIP graph of 20 back-to-back launches replayed 10 times (no host gaps), time / 200.  A least-squares
fit of t0 (the fixed cost per launch: dispatch, ramp-up to full memory-level parallelism, tail, the partial reduction) and BW.
Usage: python tools/short_sweep_fit.py [J K]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmtf_pls_amd.backend import HipBackend  # noqa: E402


def graph_time(fn, launches=20, replays=10):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(launches):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (launches * replays)          # ms per launch


def main():
    J, K = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (128, 128)
    P, M = J * K, 16
    be = HipBackend("cuda:0")
    g = torch.Generator(device="cuda:0").manual_seed(0)
    rows = [512, 1024, 2048, 4096, 8192, 16384, 32768, 65536]
    Xall = torch.randn(rows[-1], P, device="cuda:0", dtype=torch.float32, generator=g)
    Yall = torch.randn(rows[-1], M, device="cuda:0", dtype=torch.float64, generator=g)
    q = torch.randn(M, device="cuda:0", dtype=torch.float64, generator=g)
    wa = torch.randn(J, device="cuda:0", dtype=torch.float64, generator=g)
    wb = torch.randn(K, device="cuda:0", dtype=torch.float64, generator=g)
    Z = be.empty(P)
    qpart = be.empty(be.n_partials * M)
    res = {"contraction (contract_vec + reduce_rows, u = Y q inside)": [], "score (+ Y^T t partials)": [], "empty launch pair": []}
    one = be.empty(1)
    for I in rows:
        X, Y = Xall[:I], Yall[:I]
        t = be.empty(I)
        be.mode0_contract_yq(X, Y, q, False, out=Z)
        be.score_gram(X, J, K, wa, wb, None, t, Y, qpart)
        nbytes = I * P * 4
        for name, fn in (("contraction (contract_vec + reduce_rows, u = Y q inside)", lambda: be.mode0_contract_yq(X, Y, q, False, out=Z)),
                         ("score (+ Y^T t partials)", lambda: be.score_gram(X, J, K, wa, wb, None, t, Y, qpart))):
            ms = graph_time(fn)
            res[name].append((I, nbytes, ms))
            print(f"{name:58s} I={I:6d}  {ms * 1e3:8.2f} us  {nbytes / ms / 1e6:7.0f} GB/s", flush=True)
    ms0 = graph_time(lambda: be.normalize(one))
    print(f"a one-element kernel back to back inside a graph: {ms0 * 1e3:.2f} us per launch (dispatch floor)")
    for name, pts in res.items():
        if not pts:
            continue
        A = np.array([[1.0, nb] for _, nb, _ in pts])
        y = np.array([ms for _, _, ms in pts])
        (t0, inv_bw), *_ = np.linalg.lstsq(A, y, rcond=None)
        print(f"{name}: t0 = {t0 * 1e3:.1f} us per launch, asymptotic {1.0 / inv_bw / 1e6:.0f} GB/s; "
              f"at 8192 rows: {t0 * 1e3:.1f} + {8192 * P * 4 * inv_bw * 1e3:.1f} us")


if __name__ == "__main__":
    main()
