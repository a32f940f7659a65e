#!/usr/bin/env python3
"""The sharded direct iteration with its two all-reduces captured INSIDE the iteration's HIP graph
(EngineOptions.capture_collectives) against the segment-wise form (three captured segments, two eager collectives), on the
ONE-rank RCCL communicator a one-GPU box can create (VERDICT r3 "Next" #3a).

Prints one JSON line: bit-identity of loadings / q / scores after N iterations, whether the capture succeeded, and the time
per iteration of (unsharded graph replay | segment-wise | all-reduces in the graph).
Usage: python tools/collectives_in_graph.py [rows J K] [--steps N]"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmtf_pls_amd.backend import HipBackend  # noqa: E402
from cmtf_pls_amd.engine import Comm, EngineOptions, NipalsEngine  # noqa: E402
from cmtf_pls_amd.synthetic import synthetic_shard_device  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("shape", nargs="*", type=int, default=[8192, 128, 128])
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--responses", type=int, default=16)
    args = ap.parse_args()
    I, J, K = args.shape
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29547")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    be = HipBackend(dev)
    X, Y = synthetic_shard_device((I, J, K), args.responses, 10, error=0.1, seed=215, device=dev)

    def run_form(comm, capture):
        eng = NipalsEngine(be, comm, EngineOptions(capture_collectives=capture))
        run = eng.begin([X.clone()], Y.clone(), 3, coupled=False)
        run.use_graphs = True
        run.start_component(0)
        it = 0
        for _ in range(12):                                  # both q parities captured, squaring budget settled
            run.iterate(it)
            it += 1
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run.iterate(it)
            it += 1
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        out = {"ms_per_iteration": ms, "graphs": bool(run.use_graphs), "graph_error": run._graph_error,
               "collectives_in_graph": run._collectives_captured is True, "notes": list(run.notes), "n_graphs": len(run._graphs)}
        return out, (run.wA[0].clone(), run.wB[0].clone(), run.q.clone(), run.Ts[0].clone())

    plain, ref = run_form(None, False)
    seg, a = run_form(Comm(force=True), False)
    cap, b = run_form(Comm(force=True), True)
    res = {"shape": [I, J, K], "steps": args.steps, "rccl_ranks": dist.get_world_size(),
           "unsharded_graph_replay": plain, "segment_wise": seg, "collectives_in_graph": cap,
           "bit_identical_segment_vs_captured": all(torch.equal(x, y) for x, y in zip(a, b)),
           "bit_identical_sharded_vs_unsharded": all(torch.equal(x, y) for x, y in zip(a, ref)),
           "overhead_us_segment_wise": (seg["ms_per_iteration"] - plain["ms_per_iteration"]) * 1e3,
           "overhead_us_captured": (cap["ms_per_iteration"] - plain["ms_per_iteration"]) * 1e3}
    print(json.dumps(res), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
