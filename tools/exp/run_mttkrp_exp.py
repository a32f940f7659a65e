#!/usr/bin/env python3
"""What bounds the MTTKRP kernel (tools/exp/mttkrp_exp.hip)?  65536 x 128 x 128 f32, R = 10."""
import ctypes, os, torch
HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "libmttkrpexp.so"))
P_ = ctypes.c_void_p
lib.mttkrp_exp.argtypes = [ctypes.c_int, P_, ctypes.c_int64, ctypes.c_int, ctypes.c_int, P_, P_, ctypes.c_int, P_, ctypes.c_int, ctypes.c_int, P_]
I, A, B, R = 65536, 128, 128, 10
X = torch.randn(I, A * B, device="cuda:0", dtype=torch.float32)
WA = torch.randn(A, R, device="cuda:0", dtype=torch.float64)
WB = torch.randn(B, R, device="cuda:0", dtype=torch.float64)
out = torch.empty(I, R, device="cuda:0", dtype=torch.float64)
st = torch.cuda.current_stream().cuda_stream
ref = None
names = {0: "product structure (weights from LDS + mul, convert)", 1: "no weights", 2: "no convert", 3: "loads + MFMA only",
         4: "B operand shared by 2 row groups, UN 4", 5: "B operand shared by 4 row groups, UN 2", 6: "B shared by 4 row groups, UN 4",
         7: "B shared by 2 row groups, UN 8", 8: "4 row groups, no weights, no convert"}
for kind in range(9):
    for grid in (512, 1024, 2048):
        fn = lambda: lib.mttkrp_exp(kind, X.data_ptr(), I, A, B, WA.data_ptr(), WB.data_ptr(), R, out.data_ptr(), R, grid, st)
        assert fn() == 0
        fn()
        ev = []
        for _ in range(6):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); ev.append((a, b))
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in ev)
        chk = ""
        if kind in (0, 4, 5, 6, 7):
            if ref is None:
                W = (WA[:, None, :] * WB[None, :, :]).reshape(A * B, R)
                ref = X[:4096].double() @ W
            chk = f" max rel err {float(((out[:4096] - ref).abs().max() / ref.abs().max())):.1e}"
        print(f"{names[kind]:55s} grid {grid:5d}: {ts[3]:7.3f} ms {I * A * B * 4 / ts[3] / 1e6:7.1f} GB/s{chk}", flush=True)
