#!/usr/bin/env python3
"""score_contract on rows longer than one workgroup's registers (the row split over G workgroups, partial dot products exchanged
through HBM): correctness against score + mode0_contract and time per call.
Usage: python tools/split_row_time.py [I A B] [--dtype f32|f64] [--variants 41,22,...]   (variants: tuning builds only)"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmtf_pls_amd.backend import HipBackend  # noqa: E402
from kernel_bench import timeit  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("shape", nargs="*", type=int, default=[32768, 256, 256])
ap.add_argument("--dtype", default="f32")
ap.add_argument("--variants", default="")
args = ap.parse_args()
I, A, B = args.shape
P = A * B
dt = torch.float32 if args.dtype == "f32" else torch.float64
be = HipBackend("cuda:0")
g = torch.Generator(device="cuda:0").manual_seed(0)
X = torch.randn(I, P, device="cuda:0", dtype=dt, generator=g) + 0.5
wa = torch.randn(A, device="cuda:0", dtype=torch.float64, generator=g); wa /= wa.norm()
wb = torch.randn(B, device="cuda:0", dtype=torch.float64, generator=g); wb /= wb.norm()
sub = torch.randn(I, device="cuda:0", dtype=torch.float64, generator=g)
t, Z, t2, Z2, cs = be.empty(I), be.empty(P), be.empty(I), be.empty(P), be.empty(1)
xbytes = X.numel() * X.element_size()
be.score(X, A, B, wa, wb, None, t2)
t2 -= sub
be.mode0_contract(X, t2, False, out=Z2)
m1, _ = timeit(lambda: be.score(X, A, B, wa, wb, None, t2))
m2, _ = timeit(lambda: be.mode0_contract(X, t2, False, out=Z2))
be.score(X, A, B, wa, wb, None, t2)
t2 -= sub
be.mode0_contract(X, t2, False, out=Z2)
print(f"{I}x{A}x{B} {args.dtype}: score {m1:.3f} ms + mode0_contract {m2:.3f} ms = {m1 + m2:.3f} ms ({2 * xbytes / (m1 + m2) / 1e9:.2f} TB/s of two reads)", flush=True)
for v in (args.variants.split(",") if args.variants else [""]):
    if v:
        os.environ["CMTFPLS_SPLIT_VARIANT"] = v
    out = be.score_contract(X, A, B, wa, wb, None, t, Z, sub_own=sub, csum=cs)
    if out is None:
        print("variant", v, "declined", flush=True)
        continue
    torch.cuda.synchronize()
    et = float((t - t2).abs().max() / t2.abs().max())
    eZ = float((Z - Z2).abs().max() / Z2.abs().max())
    ec = float((cs[0] - t2.sum()).abs() / t2.abs().sum())
    med, best = timeit(lambda: be.score_contract(X, A, B, wa, wb, None, t, Z, sub_own=sub, csum=cs))
    print(f"  one read, variant '{v}': median {med:.3f} ms best {best:.3f} ms = {xbytes / med / 1e9:.2f} TB/s | max rel err t {et:.1e} Z {eZ:.1e} csum {ec:.1e}", flush=True)
