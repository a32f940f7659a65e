#!/usr/bin/env python3
"""The one-launch chain (memset + kernels) captured into a HIP graph and replayed on new Z contents: equal to the launch form?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cmtf_pls_amd.backend import HipBackend
be = HipBackend("cuda:0")
g = torch.Generator(device="cuda:0").manual_seed(7)
A = B = 128
Z = torch.empty(A * B, device="cuda:0", dtype=torch.float64)
wA, wB, info = be.empty(A), be.empty(B), be.zeros(2)
wA2, wB2, info2 = be.empty(A), be.empty(B), be.zeros(2)
def newZ():
    U = torch.randn(A, 3, device="cuda:0", dtype=torch.float64, generator=g)
    V = torch.randn(B, 3, device="cuda:0", dtype=torch.float64, generator=g)
    Z.copy_((U @ V.T + 0.3 * torch.randn(A, B, device="cuda:0", dtype=torch.float64, generator=g)).view(-1))
newZ()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    be.rank1(Z, A, B, wA, wB, info=info, n_squarings=9)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, capture_error_mode="thread_local"):
        be.rank1(Z, A, B, wA, wB, info=info, n_squarings=9)
    bad = 0
    for c in range(200):
        newZ()
        gr.replay()
        be.rank1(Z, A, B, wA2, wB2, info=info2, n_squarings=9, launches=True)
        torch.cuda.synchronize()
        if not (torch.equal(wA, wA2) and torch.equal(wB, wB2) and torch.equal(info, info2)):
            bad += 1
            if bad <= 5:
                print("replay", c, "chain info", info.tolist(), "launches info", info2.tolist(), "max|dwA|", (wA - wA2).abs().max().item(), flush=True)
print("graph replays of the chain:", bad, "mismatches of 200", flush=True)
