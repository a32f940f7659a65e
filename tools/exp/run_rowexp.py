#!/usr/bin/env python3
"""Run the access-pattern experiments of tools/exp/rowexp.hip on an X-sized buffer (GPU box)."""
import ctypes
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "librowexp.so"))
lib.exp_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]

I, P = 65536, 16384
X = torch.randn(I, P, device="cuda:0", dtype=torch.float32)
sink = torch.zeros(8192, device="cuda:0", dtype=torch.float32)
nbytes = I * P * 4
st = torch.cuda.current_stream().cuda_stream


def run(kind, nrows, rowvec, grid, pad, passes, label):
    def fn():
        rc = lib.exp_launch(kind, X.data_ptr(), nrows, rowvec, grid, pad, sink.data_ptr(), st)
        assert rc == 0, (kind, rc)
    for _ in range(2):
        fn()
    ev = []
    for _ in range(6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        ev.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    med = ts[len(ts) // 2]
    print(f"{label:58s} grid {grid:5d} pad {pad:6d}: {med:7.3f} ms  {passes * nbytes / med / 1e6:7.1f} GB/s", flush=True)


RV = P // 4      # 4096 vectors per 64 KB row
names = {0: "rmw row 1024x4 barrier", 1: "rmw row 1024x4", 2: "rmw row 512x8 barrier", 3: "rmw row 512x8", 4: "rmw row 256x16 barrier",
         5: "rmw row 256x16", 6: "read row 1024x4", 7: "read row 512x8", 8: "read row 256x16"}
for kind in (0, 1, 2, 3, 4, 5):
    for grid in (256, 512, 1024):
        for pad in (0, 81920):
            if pad and grid == 1024:
                continue
            run(kind, I, RV, grid, pad, 2, names[kind])
for kind in (6, 7, 8):
    for grid in (256, 512, 1024, 2048):
        run(kind, I, RV, grid, 0, 1, names[kind])
# half / quarter / double rows
for kind, rv, lab in ((9, RV // 2, "rmw half-row 1024x2 barrier"), (10, RV // 2, "rmw half-row 1024x2"), (11, RV // 4, "rmw quarter-row 1024x1 barrier"),
                      (12, RV * 2, "rmw double-row 1024x8 barrier")):
    for grid in (256, 512, 1024):
        run(kind, I * RV // rv, rv, grid, 0, 2, lab)
for kind, rv, lab in ((13, RV // 2, "read half-row 1024x2"), (14, RV * 2, "read double-row 1024x8")):
    for grid in (256, 512, 1024):
        run(kind, I * RV // rv, rv, grid, 0, 1, lab)
for kind, lab in ((20, "rmw row pipelined 1024x4"), (21, "rmw row pipelined 512x8"), (22, "rmw row pipelined 256x16")):
    for grid in (256, 512, 1024):
        run(kind, I, RV, grid, 0, 2, lab)
for kind, lab, passes in ((30, "rmw col-owner 256thr x2 RU4", 2), (31, "read col-owner 256thr x2 RU4", 1), (32, "rmw col-owner 256thr x4 RU4", 2),
                          (33, "read col-owner 256thr x4 RU4", 1), (34, "rmw col-owner 256thr x4 RU8", 2), (35, "rmw col-owner 1024thr x4 RU2", 2),
                          (36, "read col-owner 1024thr x4 RU2", 1), (37, "read col-owner 1024thr x4 RU4", 1)):
    for grid in (256, 512, 1024, 2048):
        run(kind, I, RV, grid, 0, passes, lab)
