#!/usr/bin/env python3
"""Address maps for the mode-0 contraction (tools/exp/contract_exp.hip), 65536 x 128 x 128 f32."""
import ctypes, os, torch
HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "libcontractexp.so"))
P_ = ctypes.c_void_p
lib.contract_exp.argtypes = [ctypes.c_int, P_, ctypes.c_int64, ctypes.c_int64, P_, P_, ctypes.c_int, P_]
I, P = 65536, 16384
X = torch.randn(I, P, device="cuda:0", dtype=torch.float32)
u = torch.randn(I, device="cuda:0", dtype=torch.float64)
part = torch.zeros(1024 * P, device="cuda:0", dtype=torch.float64)
st = torch.cuda.current_stream().cuda_stream
want = None
names = {0: "col owner 256thr U4 RU4 (product)", 1: "col owner 256thr U2 RU4", 2: "col owner 256thr U4 RU2", 3: "col owner 256thr U2 RU8",
         10: "row segment 1024thr NV4 RU2", 11: "row segment 1024thr NV4 RU4", 12: "half row 1024thr NV2 RU2", 13: "half row 1024thr NV2 RU4",
         14: "half row 1024thr NV2 RU8", 15: "quarter row 1024thr NV1 RU4", 16: "quarter row 1024thr NV1 RU8"}
for kind in (0, 1, 2, 3, 10, 11, 12, 13, 14, 15, 16):
    for grid in ((512, 1024, 2048) if kind < 10 else (256, 512)):
        fn = lambda: lib.contract_exp(kind, X.data_ptr(), I, P, u.data_ptr(), part.data_ptr(), grid, st)
        part.zero_()
        assert fn() == 0
        torch.cuda.synchronize()
        nrows = grid // (P // (1024 * {0: 4, 1: 2, 2: 4, 3: 2}[kind])) if kind < 10 else grid // (P // (4096 * {10: 4, 11: 4, 12: 2, 13: 2, 14: 2, 15: 1, 16: 1}[kind]))
        Z = part[: nrows * P].view(nrows, P).sum(0)
        if want is None:
            want = (X[:, :64].double().t() @ u)
        err = float((Z[:64] - want).abs().max() / want.abs().max())
        fn()
        ev = []
        for _ in range(8):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); ev.append((a, b))
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in ev)
        print(f"{names[kind]:36s} grid {grid:5d} ({nrows:4d} partial rows): {ts[4]:7.3f} ms {I * P * 4 / ts[4] / 1e6:7.1f} GB/s  err {err:.1e}", flush=True)
