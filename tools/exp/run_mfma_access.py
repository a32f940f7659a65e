#!/usr/bin/env python3
"""Does the MTTKRP's operand layout (16 rows x 64 B per load instruction) cost HBM bandwidth? (GPU box)"""
import ctypes, os, torch
HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "librowexp.so"))
lib.exp_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
I, P = 65536, 16384
X = torch.randn(I, P, device="cuda:0", dtype=torch.float32)
sink = torch.zeros(8192, device="cuda:0", dtype=torch.float32)
st = torch.cuda.current_stream().cuda_stream
for kind, lab in ((40, "16 rows x 64 B per instruction, 4 in flight"), (41, "16 rows x 64 B per instruction, 8 in flight"),
                  (42, "1 row x 1 KB per instruction, 4 in flight"), (43, "1 row x 1 KB per instruction, 8 in flight")):
    for grid in (512, 1024, 2048):
        for pad in (0, 40960):
            fn = lambda: lib.exp_launch(kind, X.data_ptr(), I, P // 4, grid, pad, sink.data_ptr(), st)
            fn(); fn()
            ev = []
            for _ in range(6):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); fn(); b.record(); ev.append((a, b))
            torch.cuda.synchronize()
            ts = sorted(a.elapsed_time(b) for a, b in ev)
            print(f"{lab:48s} grid {grid:5d} lds pad {pad:6d}: {ts[3]:7.3f} ms {I * P * 4 / ts[3] / 1e6:7.1f} GB/s", flush=True)
