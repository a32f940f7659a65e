#!/usr/bin/env python3
"""One-pass transform (MTTKRP with 64-152 KB of loadings in LDS, one workgroup per CU) against the sequential
project-and-deflate path it replaces, at trailing shapes beyond the old 96 KB limit."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmtf_pls_amd.backend import HipBackend

be = HipBackend("cuda:0")
R = 10


def ev(fn, n=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


SHAPES = [(65536, 128, 128), (16384, 256, 256), (8192, 512, 256), (4096, 512, 512), (4096, 640, 576)]
if len(sys.argv) > 1:                              # e.g. 65536x256x256,32768x512x256
    SHAPES = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1].split(",")]
for (I, A, B) in SHAPES:
    g = torch.Generator(device="cuda:0").manual_seed(1)
    X = torch.randn(I, A * B, device="cuda:0", dtype=torch.float32, generator=g)
    WA = torch.randn(A, R, device="cuda:0", dtype=torch.float64, generator=g)
    WB = torch.randn(B, R, device="cuda:0", dtype=torch.float64, generator=g)
    out = be.empty(I, R)
    lds = (A + B) * 16 * 8
    if be.mttkrp(X, A, B, WA, WB, out) is None:
        print(f"{I}x{A}x{B}: loadings {lds/1024:.0f} KB: refused (sequential path)")
        continue
    t1 = ev(lambda: be.mttkrp(X, A, B, WA, WB, out))
    t = be.empty(I)
    wa, wb = WA[:, 0].contiguous(), WB[:, 0].contiguous()
    Xc = X.clone()

    def one_component():                           # what the sequential path does per component at this shape
        if be.score_deflate(Xc, A, B, wa, wb, None, t) is None:
            be.score(Xc, A, B, wa, wb, None, t)
            be.deflate(Xc, A, B, t, wa, wb)

    t2 = ev(one_component, n=2)
    gb = X.numel() * 4 / 1e9
    print(f"{I}x{A}x{B} f32 ({gb:.2f} GB), R={R}: loadings {lds/1024:.0f} KB in LDS: mttkrp {t1:.3f} ms = {gb/t1:.2f} TB/s;"
          f" one project-and-deflate pass {t2:.3f} ms -> sequential path ~{R*t2:.1f} ms ({R*t2/t1:.1f}x)", flush=True)
    del X, Xc
