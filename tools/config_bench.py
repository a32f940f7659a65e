#!/usr/bin/env python3
"""NIPALS iterations/s and sec-to-fit for BASELINE.json configs[2] (coupled tensor + matrix block)
and configs[3] (30 % NaN) at full size on one GPU, next to configs[1].  Synthetic data formed on the
device (cmtf_pls_amd.synthetic.synthetic_shard_device).  Usage: python tools/config_bench.py [--steps 20]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmtf_pls_amd.backend import HipBackend  # noqa: E402
from cmtf_pls_amd.engine import NipalsEngine  # noqa: E402
from cmtf_pls_amd.synthetic import synthetic_shard_device  # noqa: E402


def time_iters(eng, Xs, Y, R, coupled, steps, graphs):
    run = eng.begin([x.clone() for x in Xs], Y.clone(), R, coupled=coupled)
    run.start_component(0)
    run.use_graphs = graphs
    it = 0
    for _ in range(6):
        run.iterate(it)
        it += 1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        run.iterate(it)
        it += 1
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def time_fit(eng, Xs, Y, R, coupled, algorithm):
    Xc, Yc = [x.clone() for x in Xs], Y.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st = eng.fit(Xc, Yc, R, tol=1e-8, max_iter=100, coupled=coupled, algorithm=algorithm)
    torch.cuda.synchronize()
    return time.perf_counter() - t0, list(st.n_iter)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--shape", type=int, nargs=3, default=[65536, 128, 128])
    args = ap.parse_args()
    I, J, K = args.shape
    M, R = 16, 10
    dev = torch.device("cuda:0")
    eng = NipalsEngine(HipBackend(dev), None)
    cases = {
        "configs[1] plain": dict(),
        "configs[2] coupled (+ matrix block I x 512)": dict(matrix_block=512),
        "configs[3] 30% NaN": dict(nan_fraction=0.3),
    }
    for name, kw in cases.items():
        out = synthetic_shard_device((I, J, K), M, R, error=0.1, seed=215, device=dev, dtype=torch.float32, **kw)
        Xs, Y = ([out[0], out[2]], out[1]) if len(out) == 3 else ([out[0]], out[1])
        coupled = len(Xs) > 1
        xbytes = sum(x.numel() * x.element_size() for x in Xs)
        rec = {"case": name, "X_GB": xbytes / 1e9}
        for graphs in (False, True):
            ms = time_iters(eng, Xs, Y, R, coupled, args.steps, graphs) * 1e3
            rec["ms_per_iter_graphs" if graphs else "ms_per_iter_eager"] = ms
            rec["x_TBps_graphs" if graphs else "x_TBps_eager"] = 2 * xbytes / ms / 1e9
        for algo in ("direct", "xcov"):
            s0, n = time_fit(eng, Xs, Y, R, coupled, algo)     # first call: includes one-time allocations / module loads
            s1, n = time_fit(eng, Xs, Y, R, coupled, algo)
            rec[f"fit_{algo}_s"] = s1
            rec[f"fit_{algo}_first_call_s"] = s0
            rec[f"fit_{algo}_iters"] = sum(n)
        print(json.dumps(rec), flush=True)
        del Xs, Y, out
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
