#!/usr/bin/env python3
"""Time every X sweep at a benchmark shape with HIP events on the launch stream; print achieved
GB/s against ALGORITHMIC bytes.  Usage: python tools/kernel_bench.py [I A B] [--dtype f32]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmtf_pls_amd.backend import HipBackend  # noqa: E402


def timeit(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("shape", nargs="*", type=int, default=[65536, 128, 128])
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--M", type=int, default=16)
    ap.add_argument("--only", default="all")
    ap.add_argument("--R", type=int, default=10, help="components of the MTTKRP")
    args = ap.parse_args()
    I, A, B = args.shape
    P = A * B
    dt = torch.float32 if args.dtype == "f32" else torch.float64
    es = 4 if args.dtype == "f32" else 8
    be = HipBackend("cuda:0")
    g = torch.Generator(device="cuda:0").manual_seed(0)
    X = torch.randn(I, P, device="cuda:0", dtype=dt, generator=g)
    u = torch.randn(I, device="cuda:0", dtype=torch.float64, generator=g)
    wa = torch.randn(A, device="cuda:0", dtype=torch.float64, generator=g); wa /= wa.norm()
    wb = torch.randn(B, device="cuda:0", dtype=torch.float64, generator=g); wb /= wb.norm()
    Y = torch.randn(I, args.M, device="cuda:0", dtype=torch.float64, generator=g)
    q = torch.randn(args.M, device="cuda:0", dtype=torch.float64, generator=g)
    t = be.empty(I); t2 = be.empty(I); Z = be.empty(P); un = be.empty(I)
    tsmall = torch.randn(I, device="cuda:0", dtype=torch.float64, generator=g) * 1e-3
    rowcnt = torch.full((I,), float(P), device="cuda:0", dtype=torch.float64)
    xbytes = I * P * es
    rows = []

    def rec(name, fn, nbytes):
        med, best = timeit(fn)
        rows.append((name, med, best, nbytes / med / 1e6))
        print(f"{name:28s} median {med:8.3f} ms  best {best:8.3f} ms  {nbytes/med/1e6:8.1f} GB/s (alg. bytes {nbytes/1e9:.3f} GB)", flush=True)

    if args.only == "mfma":
        S = be.empty(args.M, P)
        rec(f"xcov S=X^T Y (M={args.M}, f64 MFMA)", lambda: be.xcov(X, Y, False, out=S), xbytes)
        WAm = torch.randn(A, args.R, device="cuda:0", dtype=torch.float64, generator=g)
        WBm = torch.randn(B, args.R, device="cuda:0", dtype=torch.float64, generator=g)
        Mo = be.empty(I, args.R)
        rec(f"mttkrp X(WA.WB) (R={args.R}, f64 MFMA)", lambda: be.mttkrp(X, A, B, WAm, WBm, Mo), xbytes)
        if args.dtype == "f32":
            rec(f"xcov mixed (M={args.M}, f32 MFMA)", lambda: be.xcov(X, Y, False, out=S, mixed=True), xbytes)
            rec(f"mttkrp mixed (R={args.R}, f32 MFMA)", lambda: be.mttkrp(X, A, B, WAm, WBm, Mo, mixed=True), xbytes)
        return
    if args.only == "contract":
        rec("mode0_contract", lambda: be.mode0_contract(X, u, False, out=Z), xbytes)
        rec("mode0_contract_yq (u = Y q inside)", lambda: be.mode0_contract_yq(X, Y, q, False, out=Z), xbytes)
        rec("score", lambda: be.score(X, A, B, wa, wb, None, t), xbytes)
        return
    # reference point: a plain device copy of X (read + write)
    X2 = torch.empty_like(X)
    rec("torch copy (r+w)", lambda: X2.copy_(X), 2 * xbytes)
    del X2
    rec("mode0_contract", lambda: be.mode0_contract(X, u, False, out=Z), xbytes)
    rec("mode0_contract masked", lambda: be.mode0_contract(X, u, True, out=Z), xbytes)
    if args.M <= 64:
        rec("mode0_contract_yq (u = Y q inside)", lambda: be.mode0_contract_yq(X, Y, q, False, out=Z), xbytes)
    rec("score", lambda: be.score(X, A, B, wa, wb, None, t), xbytes)
    if args.M <= 64:
        qpart = be.empty(be.n_partials * args.M)
        rec("score_gram (+ Y^T t partials)", lambda: be.score_gram(X, A, B, wa, wb, None, t, Y, qpart), xbytes)
    rec("score masked", lambda: be.score(X, A, B, wa, wb, rowcnt, t), xbytes)
    rec("deflate (r+w)", lambda: be.deflate(X, A, B, tsmall, wa, wb), 2 * xbytes)
    if args.M <= 64:
        rec("deflate_contract_yq (r+w)", lambda: be.deflate_contract_yq(X, A, B, tsmall, wa, wb, Y, q, False, out=Z), 2 * xbytes)
    rec("center (r+w)", lambda: be.center(X, torch.zeros(P, device='cuda:0', dtype=torch.float64), False), 2 * xbytes)
    rec("score_deflate fused (r+w)", lambda: be.score_deflate(X, A, B, wa, wb, None, t2), 2 * xbytes)
    if args.only == "sweeps":
        return
    rec("colstats", lambda: be.colstats(X), xbytes)
    rec("center (r+w)", lambda: be.center(X, torch.zeros(P, device='cuda:0', dtype=torch.float64), False), 2 * xbytes)
    wA, wB = be.empty(A), be.empty(B)
    Zm = torch.randn(P, device="cuda:0", dtype=torch.float64, generator=g)
    rec("rank1 (A x B)", lambda: be.rank1(Zm, A, B, wA, wB), P * 8)
    S = be.empty(args.M, P)
    rec(f"xcov S=X^T Y (M={args.M}, f64 MFMA)", lambda: be.xcov(X, Y, False, out=S), xbytes)
    R = 10
    WAm = torch.randn(A, R, device="cuda:0", dtype=torch.float64, generator=g)
    WBm = torch.randn(B, R, device="cuda:0", dtype=torch.float64, generator=g)
    Mo = be.empty(I, R)
    rec(f"mttkrp X(WA.WB) (R={R}, f64 MFMA)", lambda: be.mttkrp(X, A, B, WAm, WBm, Mo), xbytes)
    rec("gram_tn Y^T t", lambda: be.gram_tn(Y, t), I * args.M * 8)
    rec("rowdot u = Y q (+du2)", lambda: be.rowdot(Y, q, un, u), I * args.M * 8)


if __name__ == "__main__":
    main()
