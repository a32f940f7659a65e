#!/usr/bin/env python3
"""Rank-1 extraction (tpls.py:86-88): the one-launch chain of Gram squarings (syrk_chain_kernel, round 4) against the
launch-per-squaring form -- bit equality of the loadings and time per extraction at a given budget.
Usage: python tools/rank1_chain_time.py [A B ...pairs]   (default 128 128  256 256  96 160  128 16384  200 200)"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmtf_pls_amd.backend import HipBackend  # noqa: E402
from kernel_bench import timeit  # noqa: E402

args = [int(a) for a in sys.argv[1:]] or [128, 128, 256, 256, 96, 160, 128, 16384, 200, 200, 16, 4096]
be = HipBackend("cuda:0")
rng = np.random.default_rng(0)
for A, B in zip(args[0::2], args[1::2]):
    n = min(A, B)
    # a spectrum like the benchmark recipe's: sigma_2 / sigma_1 = 0.9
    U, _ = np.linalg.qr(rng.normal(size=(A, n)))
    V, _ = np.linalg.qr(rng.normal(size=(B, n)))
    sv = 0.9 ** np.arange(n)
    Z = torch.from_numpy((U * sv) @ V.T).cuda().contiguous().view(-1)
    out = {}
    for launches in (True, False):
        wA, wB, info = be.empty(A), be.empty(B), be.empty(2)
        be.rank1(Z, A, B, wA, wB, info=info, n_squarings=30, launches=launches)
        torch.cuda.synchronize()
        out[launches] = (wA.clone(), wB.clone(), info.clone())
    same = all(torch.equal(a, b) for a, b in zip(out[True], out[False]))
    used = int(out[False][2][1].item())
    err = min(float((out[False][0].cpu() - torch.from_numpy(s * U[:, 0])).abs().max()) for s in (1.0, -1.0))
    line = f"{A}x{B}: chain == launches bitwise: {same}; converged {out[False][2][0].item():.0f}, squarings used {used}, |wA - u1| {err:.1e}"
    for budget in (used + 1, 30):
        tl, _ = timeit(lambda: be.rank1(Z, A, B, wA, wB, info=info, n_squarings=budget, launches=True), n=30, warm=5)
        tc, _ = timeit(lambda: be.rank1(Z, A, B, wA, wB, info=info, n_squarings=budget, launches=False), n=30, warm=5)
        line += f" | budget {budget}: launches {tl * 1e3:.1f} us, chain {tc * 1e3:.1f} us"
    print(line, flush=True)
