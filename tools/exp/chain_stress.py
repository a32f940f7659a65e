#!/usr/bin/env python3
"""Rank-1 extraction: the one-launch chain against the launch-per-squaring form on a DIFFERENT random Z every call (stale data of
an earlier call would show), varying budgets and sizes; bitwise comparison of loadings and info.
Usage: python tools/exp/chain_stress.py [calls] [procs]   (procs > 1: that many copies of this test at once on cuda:0)"""
import os, subprocess, sys
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 400
procs = int(sys.argv[2]) if len(sys.argv) > 2 else 1
if procs > 1:
    ps = [subprocess.Popen([sys.executable, __file__, str(calls), "1", str(i)]) for i in range(procs)]
    sys.exit(max(p.wait() for p in ps))
tag = sys.argv[3] if len(sys.argv) > 3 else "0"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cmtf_pls_amd.backend import HipBackend
be = HipBackend("cuda:0")
g = torch.Generator(device="cuda:0").manual_seed(1234 + int(tag))
bad = gave = 0
for c in range(calls):
    A, B = [(128, 128), (256, 256), (64, 48), (128, 96)][c % 4]
    n = min(A, B)
    U = torch.randn(A, 3, device="cuda:0", dtype=torch.float64, generator=g)
    V = torch.randn(B, 3, device="cuda:0", dtype=torch.float64, generator=g)
    Z = (U @ V.T + 0.3 * torch.randn(A, B, device="cuda:0", dtype=torch.float64, generator=g)).contiguous().view(-1)
    budget = [30, 7, 9, 5, 12][c % 5]
    outs = []
    for launches in (True, False):
        wA, wB, info = be.empty(A), be.empty(B), be.zeros(2)
        be.rank1(Z, A, B, wA, wB, info=info, n_squarings=budget, launches=launches)
        outs.append((wA, wB, info))
    torch.cuda.synchronize()
    if outs[1][2][1].item() < 0:
        gave += 1
        continue
    same = all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))
    if not same:
        bad += 1
        if bad <= 5:
            print(f"[{tag}] call {c} {A}x{B} budget {budget}: launches info {outs[0][2].tolist()} chain info {outs[1][2].tolist()} max|dwA| {(outs[0][0]-outs[1][0]).abs().max().item():.3e}", flush=True)
print(f"[{tag}] {calls} calls: {bad} mismatches, {gave} give-ups (chain enabled at the end: {be.lib.cmtfpls_rank1_chain_enabled()})", flush=True)
sys.exit(1 if bad else 0)
