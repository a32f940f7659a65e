#!/usr/bin/env python3
"""Phase stamps of syrk_chain_kernel's workgroup (0, 0), every step (a library built with -DCMTFPLS_CHAIN_PROFILE, see the build lines in
HISTORY section 9): CMTFPLS_LIB=tools/exp/prof_lib/libcmtfpls_prof.so python tools/exp/chain_profile.py [A B]"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cmtf_pls_amd import _lib
_lib.SIGNATURES["cmtfpls_debug_chain_prof"] = (ctypes.c_int, [ctypes.c_void_p])
from cmtf_pls_amd.backend import HipBackend
A, B = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (128, 128)
be = HipBackend("cuda:0")
rng = np.random.default_rng(0)
n = min(A, B)
U, _ = np.linalg.qr(rng.normal(size=(A, n))); V, _ = np.linalg.qr(rng.normal(size=(B, n)))
Z = torch.from_numpy((U * (0.9 ** np.arange(n))) @ V.T).cuda().contiguous().view(-1)
wA, wB, info = be.empty(A), be.empty(B), be.zeros(2)
for _ in range(5):
    be.rank1(Z, A, B, wA, wB, info=info, n_squarings=9)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 1024)()
assert be.lib.cmtfpls_debug_chain_prof(buf) == 0
st = np.array(buf[:], dtype=np.int64).reshape(128, 8)
names = ["top", "polled", "staged(barrier)", "mfma done", "reduced(barrier)", "C stored", "fro/tr stored"]
used = int(info[1].item())
print(f"{A}x{B}: squarings used {used}; microseconds since the step's top (100 MHz clock), polls repeated")
for s in range(used + 2):
    t = st[s, :7]
    if t[0] == 0:
        break
    rel = (t - t[0]) / 100.0
    nxt = (st[s + 1, 0] - t[0]) / 100.0 if st[s + 1, 0] else float("nan")
    print(f"step {s}: " + "  ".join(f"{nm} {r:5.2f}" for nm, r in zip(names[1:], rel[1:])) + f"  | next step's top {nxt:5.2f} | repeats {st[s, 7]}")
