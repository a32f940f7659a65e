#!/usr/bin/env python3
"""bench.py -- NIPALS iterations/sec (+ sec-to-fit R=10) of the MI355X tPLS engine.

Contract: ``python bench.py --gpus N --steps K --warmup W``.  With N > 1 and no WORLD_SIZE in the
environment this process touches no GPU: it starts ``python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same flags>`` as a child (one rank per GPU over
RCCL), forwards its output and exits with its code.  Launched BY torch.distributed.run (WORLD_SIZE set),
it is one rank; WORLD_SIZE != --gpus is refused with a non-zero exit before any GPU call, and the process
group is checked to have exactly N ranks.

A *step* is ONE direct-form NIPALS inner iteration of the product engine
(cmtf_pls_amd.fitrun.FitRun.iterate = reference tpls.py:80-107): mode-0 contraction (one read of X),
rank-1 extraction, score contraction (second read of X), Y update, convergence norm read back to the
host exactly as the reference tests it every iteration.  Inputs are resident in HBM before the timed
region.  Workload: BASELINE.json configs[1] (X 65536x128x128 f32, Y 65536x16, R=10), STRONG scaling:
the same X is row-sharded over the N ranks, as the metric is quoted ("iters/sec on X 65536x128x128 at
1/2/4/8 GPU"); per iteration the ranks all-reduce Z (J*K doubles) and Y^T t (M doubles).

Timing (round 3): W warm-up steps, then ``--repeats`` (default 5) windows of EXACTLY K steps each, every window bracketed by a
barrier + ``torch.cuda.synchronize()`` on both sides and reduced with MAX over ranks; ``value`` = K / the MEDIAN window
(``value_min`` / ``value_max`` = the slowest / fastest window).  At N = 1 with the default workload the line also carries
``north_star``: BASELINE configs[4] (X 262144x256x256 f32 = 68.7 GB, Y 262144x32) formed on the device after the cfg-2 legs,
direct iterations under graph replay and the component's deflation sweeps (skipped when less than 150 GB of HBM are free).

Rank 0 prints ONE JSON line; ``roofline`` is the dominant kernel (the mode-0 contraction launch) from HIP
events on the launch stream inside the timed region, against the 8 TB/s spec peak AND against streaming
ceilings measured in the same run (``peak_measured_read`` / ``peak_measured_rmw``: plain float4
read-only / read-modify-write kernels of libcmtfpls over an X-sized buffer); ``roofline.traffic`` = HBM bytes per launch from
profiles/pmc_traffic.json (separate rocprofv3 --pmc passes of this command), quoted only while the kernel it names is in the
running library and the sources it was compiled from are unchanged (else null, with the reason); ``cpu_baseline`` is the
NumPy oracle's inner loop timed on this host on a row sample (rank 0, N=1 only).  ``fit`` is the sec-to-fit leg: whole
fits (default tol / max_iter, preprocessing included) through the direct loop, the cross-covariance form and its
f32-MFMA variant, each run twice (``first_call_seconds``, ``seconds``).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
F64_MFMA_PEAK_TF = 78.6  # MI355X dense f64 matrix peak (v_mfma_f64_16x16x4_f64)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--shape", type=int, nargs=3, default=[65536, 128, 128], help="I J K of the whole job")
    ap.add_argument("--responses", type=int, default=16)
    ap.add_argument("--components", type=int, default=10)
    ap.add_argument("--noise", type=float, default=0.1)
    ap.add_argument("--no-fit", action="store_true", help="skip the sec-to-fit leg")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-ceilings", action="store_true", help="skip the measured streaming ceilings")
    ap.add_argument("--cpu-rows", type=int, default=2048)
    ap.add_argument("--graphs", type=int, default=1, help="replay the iteration's launch sequence as a HIP graph (0 = eager)")
    ap.add_argument("--repeats", type=int, default=5, help="timed windows of exactly --steps iterations each; value = median")
    ap.add_argument("--no-north-star", action="store_true", help="skip the 262144x256x256 leg (BASELINE.json north_star)")
    ap.add_argument("--north-star-steps", type=int, default=10)
    ap.add_argument("--rank1-chain", type=int, default=1,
                    help="0: the launch-per-squaring form of the rank-1 extraction instead of the one-launch chain (A/B)")
    ap.add_argument("--capture-collectives", type=int, default=0,
                    help="N > 1: capture the two per-iteration all-reduces inside the iteration's HIP graph (EngineOptions.capture_collectives; "
                         "falls back to the segment-wise form when the capture fails)")
    return ap.parse_args(argv)


# ---- launcher (no GPU call may happen before or inside these two functions) -----------------------
def launcher_command(n_gpus, argv, port=None):
    """The child command that runs this file as n_gpus ranks (one per GPU) of one node."""
    if port is None:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def resolve_world(args, environ):
    """('launch', N) when this process must start its own ranks, ('rank', world) when it is one rank.
    A WORLD_SIZE that disagrees with --gpus is refused (SystemExit, non-zero) -- a scaling run must never
    record N=1 work under n_gpus=N."""
    if args.gpus < 1:
        raise SystemExit(f"--gpus must be >= 1, got {args.gpus}")
    ws = environ.get("WORLD_SIZE")
    if ws is None:
        return ("launch", args.gpus) if args.gpus > 1 else ("rank", 1)
    if int(ws) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={ws}: refusing to run a mislabelled job")
    return ("rank", int(ws))


def synth_shard(I_total, J, K, M, L, noise, row0, rows, device, seed=215):
    """Reference recipe (synthetic.py:59-74) with the dense shard formed on the GPU
    (cmtf_pls_amd.synthetic.synthetic_shard_device): X fp32, Y f64."""
    from cmtf_pls_amd.synthetic import synthetic_shard_device
    return synthetic_shard_device((I_total, J, K), M, L, error=noise, seed=seed, row0=row0, rows=rows, device=device)


class EventTimer:
    """Brackets backend calls with HIP events on the current (launch) stream."""

    def __init__(self, be, names):
        self.be, self.records, self.on, self.only = be, {n: [] for n in names}, False, None
        for n in names:
            setattr(be, n, self._wrap(n, getattr(be, n)))

    def _wrap(self, name, fn):
        def wrapped(*a, **k):
            if not self.on or (self.only is not None and name not in self.only):
                return fn(*a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **k)
            e1.record()
            self.records[name].append((e0, e1))
            return out
        return wrapped

    def mean_ms(self, name):
        ev = self.records[name]
        return sum(a.elapsed_time(b) for a, b in ev) / max(len(ev), 1)

    def time_calls(self, fn, n=5, warm=1):
        """Average duration (ms) of n calls of fn, each bracketed by its own event pair, after `warm` untimed calls."""
        for _ in range(warm):
            fn()
        pairs = []
        for _ in range(n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            pairs.append((e0, e1))
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in pairs) / n


def cpu_baseline(J, K, M, L, noise, rows, I_total):
    """The oracle's inner loop (NumPy float64, the reference's arithmetic) on a row sample."""
    import oracle as O
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    x, y, _ = O.import_synthetic((rows, J, K), M, L, error=noise)
    xc, yc = x - x.mean(0), y - y.mean(0)
    O.nipals_inner_loop(xc, yc, 1)
    n, t0 = 0, time.perf_counter()
    while True:
        O.nipals_inner_loop(xc, yc, 2)
        n += 2
        dt = time.perf_counter() - t0
        if dt > 12.0 or n >= 200:
            break
    per_iter_sample = dt / n
    per_iter_full = per_iter_sample * (I_total / rows)
    return {"value": 1.0 / per_iter_full, "unit": "it/s", "cores": int(threads), "kind": "port",
            "sample": f"oracle.nipals_inner_loop (NumPy f64) on {rows} of {I_total} rows x {J}x{K}, {n} iterations in "
                      f"{dt:.1f} s = {per_iter_sample*1e3:.1f} ms/iter on the sample, scaled linearly in rows",
            "host_cpu_count": os.cpu_count()}


def pmc_traffic_for(dom):
    """(HBM bytes per launch of the dominant kernel, provenance) from profiles/pmc_traffic.json -- separate rocprofv3 --pmc
    passes of an earlier run of this command, not live counters.  REFUSED (None, reason) when it cannot describe the
    kernels in the library that is running now: the profile must name the kernel, the kernel must still be in
    libcmtfpls.so, and the source file it was compiled from must hash to what the profile recorded."""
    import hashlib
    pmc_file = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(pmc_file):
        return None, None
    doc = json.load(open(pmc_file))
    key = {"mode0_contract": "contract_vec_kernel", "score": "score_kernel"}[dom]
    if key not in doc.get("kernels", {}):
        return None, f"profiles/pmc_traffic.json has no entry for {key}"
    from cmtf_pls_amd import _lib
    try:
        in_lib = key.encode() in open(_lib.LIB_PATH, "rb").read()
    except OSError:
        in_lib = False
    if not in_lib:
        return None, f"refused: {key} of profiles/pmc_traffic.json is not a kernel of the current libcmtfpls.so"
    for rel, want in (doc.get("source_sha256") or {}).items():
        path = os.path.join(ROOT, rel)
        have = hashlib.sha256(open(path, "rb").read()).hexdigest() if os.path.exists(path) else None
        if have != want:
            return None, f"refused: {rel} changed since profiles/pmc_traffic.json ({doc.get('round')}) was taken; re-run tools/pmc_traffic.py"
    if not doc.get("source_sha256"):
        return None, "refused: profiles/pmc_traffic.json records no source hashes (taken before round 3)"
    return (doc["kernels"][key]["hbm_bytes_per_launch"],
            f"profiles/pmc_traffic.json ({doc.get('round', '?')}; separate rocprofv3 --pmc passes of this command, sources unchanged since)")


def north_star_leg(be, eng_cls, device, steps, repeats, timer):
    """BASELINE.json north_star workload on ONE GPU: X 262144 x 256 x 256 f32 (68.7 GB), Y 262144 x 32, R = 10 -- direct
    NIPALS iterations (two full reads of X each) replayed as a HIP graph, then the component's deflation sweeps.  Returns
    None when less than 150 GB of HBM are free."""
    from cmtf_pls_amd.synthetic import synthetic_shard_device
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    if free < 150e9:
        return {"skipped": f"{free / 1e9:.0f} GB of HBM free, 150 GB needed"}
    I, J, K, M, R = 262144, 256, 256, 32, 10
    X, Y = synthetic_shard_device((I, J, K), M, R, error=0.1, seed=215, device=device)
    eng = eng_cls(be, None)
    t0 = time.perf_counter()
    run = eng.begin([X], Y, R, coupled=False)          # centres X in place: no second copy of the 68.7 GB
    torch.cuda.synchronize()
    pre_s = time.perf_counter() - t0
    run.start_component(0)
    it = 0
    for _ in range(3):
        run.iterate(it)
        it += 1
    xbytes = I * J * K * X.element_size()
    # eager windows with events (per-sweep rates), then graph-replay windows (the it/s figure)
    for n in timer.records:
        timer.records[n] = []
    timer.on = True
    for _ in range(max(2, steps // 2)):
        run.iterate(it)
        it += 1
    torch.cuda.synchronize()
    timer.on = False
    sweeps = {}
    for name, src in (("contraction", "mode0_contract_yq"), ("score", "score_gram")):
        if timer.records[src]:
            ms = timer.mean_ms(src)
            sweeps[name] = {"ms": ms, "GBps": xbytes / ms / 1e6, "frac": xbytes / ms / 1e6 / HBM_PEAK_GBPS}
    run.use_graphs = True
    for _ in range(4):
        run.iterate(it)
        it += 1
    wins = []
    for _ in range(max(1, repeats)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            run.iterate(it)
            it += 1
        torch.cuda.synchronize()
        wins.append(time.perf_counter() - t0)
    med = float(np.median(wins))
    # the deflation of the component: the fit's own fused form (X -= t w^T, then the next component's first contraction
    # from the same registers) through finish_component, and the plain sweep; 2 * I * P * s algorithmic bytes each
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run.finish_component(0)
    e1.record()
    torch.cuda.synchronize()
    finish_ms = e0.elapsed_time(e1)
    blk = run.blocks[0]
    tz = torch.zeros(I, dtype=torch.float64, device=device)
    Zs = torch.empty(J * K, dtype=torch.float64, device=device)
    q1 = torch.ones(M, dtype=torch.float64, device=device)
    defl = {}
    for name, fn in (("deflate", lambda: be.deflate(run.X2[0], blk.A, blk.B, tz, run.wA[0], run.wB[0])),
                     ("deflate_contract_yq", lambda: be.deflate_contract_yq(run.X2[0], blk.A, blk.B, tz, run.wA[0], run.wB[0], run.Y, q1, False, out=Zs))):
        ms = timer.time_calls(fn, n=3, warm=1)
        defl[name] = {"ms": ms, "GBps": 2 * xbytes / ms / 1e6, "frac": 2 * xbytes / ms / 1e6 / HBM_PEAK_GBPS}
    out = {"workload": "BASELINE configs[4] on ONE GPU: tPLS direct NIPALS iteration, X 262144x256x256 f32 (68.7 GB, f64 accumulation), "
                       "Y 262144x32, R=10, noise 0.1", "it_per_s": steps / med, "ms_per_step": med / steps * 1e3,
           "it_per_s_min": steps / max(wins), "it_per_s_max": steps / min(wins), "steps": steps, "repeats": len(wins),
           "hip_graphs": bool(run.use_graphs), "x_reads_per_step": 2, "alg_GB_per_step": 2 * xbytes / 1e9,
           "end_to_end_GBps": 2 * xbytes / (med / steps) / 1e9, "sweeps": sweeps, "deflation": defl,
           "deflation_frac_of_hbm_peak": defl["deflate"]["frac"], "finish_component_ms": finish_ms, "preprocess_s": pre_s,
           "target": ">= 50 it/s, >= 40 % of the HBM roofline on the deflation (BASELINE.json north_star)"}
    # sec-to-fit at this shape through the cross-covariance form, on the caller's UNCENTRED tensor (never written, never
    # copied): the same synthetic tensor formed again (the legs above centred and deflated it in place).  Default tol / max_iter.
    del run, X, Y, tz, Zs
    torch.cuda.empty_cache()
    X, Y = synthetic_shard_device((I, J, K), M, R, error=0.1, seed=215, device=device)
    secs = []
    for _ in range(2):
        Yf = Y.clone()                                  # the engine centres and deflates the responses it is handed in place
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sx = eng.fit([X], Yf, R, tol=1e-8, max_iter=100, coupled=False, algorithm="xcov")
        torch.cuda.synchronize()
        secs.append(time.perf_counter() - t0)
        rep = sx.report
        if rep.get("x_written") or not rep.get("raw"):  # X was centred / deflated in place: a second fit would see other data
            secs.append(secs[0])
            break
    out["fit_xcov"] = {"seconds": secs[1], "first_call_seconds": secs[0], "n_iter": list(sx.n_iter),
                       "R2X_final": float(sx.blocks[0].r2x[-1]), "R2Y_final": float(sx.r2y[-1]),
                       "x_reads_in_all": ((R + 2) if rep.get("one_read") else 2 * R + 1) - (1 if rep.get("stats_with_s") else 0),
                       "floor_s_at_the_read_ceiling": (((R + 2) if rep.get("one_read") else 2 * R + 1) - (1 if rep.get("stats_with_s") else 0)) * xbytes / 6.95e12,
                       "path": {k: rep.get(k) for k in ("algorithm", "raw", "x_copy", "x_written", "one_read", "stats_with_s", "x_passes_per_component",
                                                        "pipelined", "declined")}}
    del X, Y, Yf, sx
    torch.cuda.empty_cache()
    return out


def measure_ceilings(be, timer, scratch, src, row_bytes):
    """Streaming ceilings of THIS box, measured with the plain float4 kernels of libcmtfpls (csrc/ceiling.hip)
    over an X-sized buffer: best of a few (address map, grid) variants per access pattern, all reported."""
    nbytes = scratch.numel() * scratch.element_size()
    res = {"buffer_GB": nbytes / 1e9, "variants": {}}
    best = {}
    maps = {0: "flat", 1: "chunk", 2: "rowwg", 3: "colowner"}
    for op, passes in (("read", 1), ("rmw", 2), ("copy", 2)):
        for mp, blocks_list in ((0, (2048, 4096)), (1, (1024, 4096)), (2, (256, 512)), (3, (256, 512, 1024))):
            for blocks in blocks_list:
                if op == "copy":
                    fn = lambda: be.ceiling("copy", src, row_bytes, blocks, dst=scratch, map=mp)
                else:
                    fn = lambda: be.ceiling(op, scratch, row_bytes, blocks, map=mp)
                if not fn():                                    # this map does not take the shape
                    continue
                ms = timer.time_calls(fn, n=4, warm=1)          # an even number of rmw launches in all: data restored
                gbps = passes * nbytes / ms / 1e6
                res["variants"][f"{op}:{maps[mp]}:{blocks}"] = round(gbps, 1)
                if gbps > best.get(op, 0.0):
                    best[op] = gbps
    res["read_GBps"], res["rmw_GBps"], res["copy_GBps"] = best["read"], best["rmw"], best["copy"]
    return res


def main():
    args = parse()
    mode, world = resolve_world(args, os.environ)
    if mode == "launch":
        # nothing above touched the GPU: starting the ranks as children is allowed
        proc = subprocess.run(launcher_command(world, sys.argv[1:]))
        raise SystemExit(proc.returncode)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("BENCH_DUMP_AFTER"):                  # rehearsals: every thread's stack after N seconds, then exit (a hung rank says where)
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["BENCH_DUMP_AFTER"]), exit=True)
    # rehearsal hooks (one-GPU box): BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and BENCH_BACKEND=gloo
    # replaces RCCL, which refuses two ranks on one device; BENCH_FORCE_DIST=1 makes a single rank create
    # its RCCL communicator and issue the engine's collectives anyway; the driver's runs use none of them
    dev_index = 0 if os.environ.get("BENCH_ONE_DEVICE") == "1" else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    import torch.distributed as dist
    force_dist = world == 1 and os.environ.get("BENCH_FORCE_DIST") == "1"
    backend = None
    if world > 1 or force_dist:
        backend = os.environ.get("BENCH_BACKEND", "nccl")
        if force_dist:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29541")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: process group has {dist.get_world_size()} ranks, --gpus {args.gpus}")

    from cmtf_pls_amd.backend import HipBackend
    from cmtf_pls_amd.engine import Comm, EngineOptions, NipalsEngine

    class TimedComm(Comm):
        """Comm whose all-reduces are bracketed by HIP events on the launch stream while `on` is set (the
        collective runs on RCCL's stream; the launch stream waits for it, so the pair spans it)."""
        on = False
        pairs = []

        def allreduce(self, t):
            if not (self.on and self.sharded):
                return super().allreduce(t)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = super().allreduce(t)
            e1.record()
            self.pairs.append((e0, e1, t.numel() * t.element_size()))
            return out

    I_total, J, K = args.shape
    M, R = args.responses, args.components
    assert I_total % world == 0
    rows = I_total // world
    X, Y = synth_shard(I_total, J, K, M, R, args.noise, rank * rows, rows, device)
    if not args.rank1_chain:
        from cmtf_pls_amd import _lib
        _lib.load().cmtfpls_rank1_chain_enable(0)
    be = HipBackend(device)
    # the product iteration calls the fused forms (u = Y q inside the contraction, Y^T t inside the score);
    # the unfused names stay bracketed for shapes that fall back to them
    timer = EventTimer(be, ["mode0_contract", "mode0_contract_yq", "score", "score_gram", "rank1", "q_update", "gram_tn",
                            "rowdot", "deflate"])
    comm = TimedComm(force=force_dist) if backend else None
    eng = NipalsEngine(be, comm, EngineOptions(capture_collectives=bool(args.capture_collectives)))

    # ---- timed leg: K direct NIPALS iterations of component 0 -------------------------------
    Xw, Yw = X.clone(), Y.clone()
    run = eng.begin([Xw], Yw, R, coupled=False)
    run.start_component(0)
    state = {"it": 0}

    def timed(steps, events):
        """barrier + sync, `steps` product iterations, sync + barrier; max over ranks (seconds)."""
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        timer.on = events
        if comm is not None:
            comm.on = events
        t0 = time.perf_counter()
        for _ in range(steps):
            run.iterate(state["it"])               # it > 0 after warm-up: convergence norm computed and read back
            state["it"] += 1
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        timer.on = False
        if comm is not None:
            comm.on = False
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        if world > 1:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        return float(tmax.item())

    # (1) eager passes with HIP events bracketing every kernel launch (roofline figures come from here;
    #     events cannot bracket kernels inside a graph replay).  `--repeats` windows of EXACTLY `--steps` iterations
    #     each; every window contributes its own per-kernel mean, the report carries median / min / max over windows
    reps = max(1, args.repeats)
    for _ in range(args.warmup):
        run.iterate(state["it"])
        state["it"] += 1
    eager_windows, kernel_windows = [], {n: [] for n in timer.records}
    for _ in range(reps):
        for n in timer.records:
            timer.records[n] = []
        eager_windows.append(timed(args.steps, events=True))
        for n in timer.records:
            if timer.records[n]:
                kernel_windows[n].append(timer.mean_ms(n))
    eager_elapsed = float(np.median(eager_windows))
    elapsed, windows = eager_elapsed, eager_windows
    # (2) the same windows with the launch sequence replayed as a HIP graph (the headline when
    #     capture works: identical kernels, ~1 host call per iteration instead of ~20 launches)
    if args.graphs:
        try:
            run.use_graphs = True
            for _ in range(max(args.warmup, 4)):   # captures both u-buffer parities
                run.iterate(state["it"])
                state["it"] += 1
            if run.use_graphs:
                windows = [timed(args.steps, events=False) for _ in range(reps)]
                elapsed = float(np.median(windows))
        except Exception as e:                     # keep the eager figure rather than lose the run
            run.use_graphs = False
            run._graph_error = repr(e)
            elapsed, windows = eager_elapsed, eager_windows
    graphs_used, graph_error = run.use_graphs, run._graph_error
    collectives_in_graph = run._collectives_captured is True
    iter_notes = list(run.notes)
    run.finish_component(0)                        # exercises the deflation sweep once (timed below by events)

    es = X.element_size()
    xbytes = rows * J * K * es                     # ALGORITHMIC bytes of one X read on this rank
    kern = {}
    # "mode0_contract" / "score" report whichever form the iteration launched (fused when its events exist)
    fused = bool(kernel_windows["mode0_contract_yq"]) and bool(kernel_windows["score_gram"])
    for name, src, nbytes in (("mode0_contract", "mode0_contract_yq" if fused else "mode0_contract", xbytes),
                              ("score", "score_gram" if fused else "score", xbytes)):
        w = kernel_windows[src]
        ms = float(np.median(w))
        kern[name] = {"ms": ms, "ms_min": min(w), "ms_max": max(w), "windows": len(w), "alg_GB": nbytes / 1e9,
                      "GBps": nbytes / ms / 1e6 if ms else None, "entry": src}
    for name in (("rank1", "q_update") if fused else ("rank1", "gram_tn", "rowdot")):
        kern[name] = {"ms": float(np.median(kernel_windows[name])) if kernel_windows[name] else None}
    # the read-modify-write sweeps of the fit path (2 * I * P * s algorithmic bytes each), HIP events over 5 launches
    blk = run.blocks[0]
    tz = torch.zeros(rows, dtype=torch.float64, device=device)
    zmean = torch.zeros(J * K, dtype=torch.float64, device=device)
    Zs = torch.empty(J * K, dtype=torch.float64, device=device)
    q1 = torch.ones(M, dtype=torch.float64, device=device)
    t_out = torch.empty(rows, dtype=torch.float64, device=device)
    rmw = {   # t = 0 / mean = 0: X keeps its values, the traffic is the real one
        "deflate": lambda: be.deflate(run.X2[0], blk.A, blk.B, tz, run.wA[0], run.wB[0]),
        "deflate_contract_yq": lambda: be.deflate_contract_yq(run.X2[0], blk.A, blk.B, tz, run.wA[0], run.wB[0], run.Y, q1, False, out=Zs),
        "score_deflate": lambda: be.score_deflate(run.X2[0], blk.A, blk.B, run.wA[0], run.wB[0], None, t_out),
        "center": lambda: be.center(run.X2[0], zmean, False),
    }
    for name, fn in rmw.items():
        try:
            ms = timer.time_calls(fn, n=5, warm=1)
            kern[name] = {"ms": ms, "alg_GB": 2 * xbytes / 1e9, "GBps": 2 * xbytes / ms / 1e6}
        except Exception as e:                     # a shape outside one fused form must not lose the run
            kern[name] = {"error": repr(e)}
    dom = "mode0_contract" if kern["mode0_contract"]["ms"] >= kern["score"]["ms"] else "score"

    ceilings = None
    if not args.no_ceilings:
        ceilings = measure_ceilings(be, timer, run.X2[0], X, J * K * es)
        for name in ("mode0_contract", "score"):
            kern[name]["frac_of_measured_read"] = kern[name]["GBps"] / ceilings["read_GBps"]
        for name in rmw:
            if "GBps" in kern[name]:
                kern[name]["frac_of_measured_rmw"] = kern[name]["GBps"] / ceilings["rmw_GBps"]

    traffic, traffic_src = pmc_traffic_for(dom) if (I_total, J, K, world) == (65536, 128, 128, 1) else (None, None)
    roofline = {"kernel": dom, "bound": "hbm", "achieved": kern[dom]["GBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": kern[dom]["GBps"] / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                "alg_bytes_per_launch": xbytes, "avg_launch_ms": kern[dom]["ms"],
                "avg_launch_ms_is": f"median over {kern[dom]['windows']} windows of the per-window mean (HIP events, launch stream)",
                "avg_launch_ms_min_max": [kern[dom]["ms_min"], kern[dom]["ms_max"]],
                "peak_measured_read": ceilings["read_GBps"] if ceilings else None,
                "peak_measured_rmw": ceilings["rmw_GBps"] if ceilings else None,
                "peak_measured_copy": ceilings["copy_GBps"] if ceilings else None,
                "frac_of_measured_read": kern[dom]["GBps"] / ceilings["read_GBps"] if ceilings else None,
                "ceilings": ceilings}

    # the two per-iteration collectives (Z: J*K doubles, Y^T t: M doubles) of the eager pass
    comm_info = None
    if comm is not None:
        by_size = {}
        for e0, e1, nb in comm.pairs:
            by_size.setdefault(nb, []).append(e0.elapsed_time(e1))
        comm_info = {"backend": backend, "ranks": dist.get_world_size(), "forced_single_rank": force_dist,
                     "in_graph": collectives_in_graph, "notes": iter_notes,
                     "allreduce_ms_by_bytes": {str(nb): sum(v) / len(v) for nb, v in sorted(by_size.items())},
                     "allreduce_ms_per_step": sum(sum(v) for v in by_size.values()) / max(args.steps * reps, 1),
                     "collectives_per_step": len(comm.pairs) / max(args.steps * reps, 1)}
    # N > 1: what every rank spent where (a scaling record must explain itself): the two sweeps, the rank-1 chain, the Y
    # update and the all-reduces of the eager windows (HIP events), next to the rank's own graph-replay step
    per_rank = None
    if world > 1:
        mine = {"rank": rank, "rows": rows, "contraction_ms": kern["mode0_contract"]["ms"], "score_ms": kern["score"]["ms"],
                "rank1_ms": kern["rank1"]["ms"], "q_update_ms": (kern.get("q_update") or {}).get("ms"),
                "allreduce_ms_per_step": comm_info["allreduce_ms_per_step"] if comm_info else None,
                "eager_ms_per_step": eager_elapsed / args.steps * 1e3, "ms_per_step": elapsed / args.steps * 1e3}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    del Xw, Yw, run

    # ---- sec-to-fit leg (default tol / max_iter, like the reference's fit()) ----------------
    fit_info = None
    mfma = None
    if not args.no_fit:
        def timed_fit(**kw):
            """One whole fit on a fresh copy of the resident X (the clone is outside the timed region; the fit's own
            preprocessing -- column statistics, centring -- is inside).  Returns (state, seconds)."""
            Xf, Yf = X.clone(), Y.clone()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            t1 = time.perf_counter()
            state = eng.fit([Xf], Yf, R, tol=1e-8, max_iter=100, coupled=False, **kw)
            torch.cuda.synchronize()
            return state, time.perf_counter() - t1

        # every fit is run twice: "first_call_seconds" includes what a process pays once (code objects of kernels not used
        # by the timed steps above, workspaces growing to size), "seconds" is the second call
        st, fit_first = timed_fit()
        st, fit_s = timed_fit()
        fit_info = {"seconds": fit_s, "first_call_seconds": fit_first, "n_iter": list(st.n_iter),
                    "iters_per_sec_in_fit": sum(st.n_iter) / fit_s,
                    "R2X_final": float(st.blocks[0].r2x[-1]), "R2Y_final": float(st.r2y[-1]), "tol": 1e-8, "max_iter": 100}
        # the same fit through the cross-covariance form (algorithm="xcov": S = X^T Y on the f64 matrix cores for the
        # first component and carried by down-dates afterwards, inner loop on S; exact re-association; X is read twice
        # per component and, without missing values, never written after the centring)
        sx, xs_first = timed_fit(algorithm="xcov")
        sx, xs = timed_fit(algorithm="xcov")
        fit_info["xcov"] = {"seconds": xs, "first_call_seconds": xs_first, "n_iter": list(sx.n_iter),
                            "iters_per_sec_in_fit": sum(sx.n_iter) / xs,
                            "R2X_final": float(sx.blocks[0].r2x[-1]), "R2Y_final": float(sx.r2y[-1]),
                            "max_abs_dT_vs_direct": float((sx.T - st.T).abs().max())}
        # which forms ran (fitrun.FitRun.build_report): algorithm after fallbacks, passes over X, centred copy or the caller's
        # tensor, pipelined, graph replay -- and every fast form the shape declined, in words
        fit_info["path"] = {"direct": st.report, "xcov": sx.report}
        # opt-in mixed precision of the S build (f32 MFMA, csrc/mixed.hip): reported, never the headline
        sm, xm_first = timed_fit(algorithm="xcov", mixed=True)
        sm, xm = timed_fit(algorithm="xcov", mixed=True)
        fit_info["xcov_mixed_f32mfma"] = {"seconds": xm, "first_call_seconds": xm_first, "n_iter": list(sm.n_iter),
                                          "max_abs_dT_vs_direct": float((sm.T - st.T).abs().max())}
        # the two matrix-core kernels, WARM (workspaces sized, 1 untimed + 5 timed launches each): achieved HBM rate
        # and matrix-pipe utilisation = flops / (time x 78.6 TF), the north star's "MFMA utilisation on the contraction"
        X2 = X.view(rows, -1)
        S = torch.empty(M, J * K, dtype=torch.float64, device=device)
        Mo = torch.empty(rows, R, dtype=torch.float64, device=device)
        WA, WB = st.blocks[0].loadings[0].contiguous(), st.blocks[0].loadings[1].contiguous()
        mfma = {"peak_TF": F64_MFMA_PEAK_TF, "note": "f64 matrix cores (v_mfma_f64_16x16x4_f64); utilisation = flops / (avg launch time x nominal peak at "
                                                      "2.4 GHz); SQ-counter view at the clock the chip holds under this load: profiles/r02y_mfma_utilisation.json"}
        for name, fn, flops in (("xcov", lambda: be.xcov(X2, Y, False, out=S), 2.0 * rows * J * K * M),
                                ("mttkrp", lambda: be.mttkrp(X2, J, K, WA, WB, Mo), 2.0 * rows * J * K * 16 * ((R + 15) // 16))):
            ms = timer.time_calls(fn, n=5, warm=1)
            mfma[name] = {"ms": ms, "alg_GB": xbytes / 1e9, "GBps": xbytes / ms / 1e6, "TFLOPs": flops / ms / 1e9,
                          "mfma_utilisation": flops / ms / 1e9 / F64_MFMA_PEAK_TF,
                          "frac_of_measured_read": xbytes / ms / 1e6 / ceilings["read_GBps"] if ceilings else None}
        # the tile-based figures above count every MFMA issued; the USEFUL share is R / 16ceil(R/16) of the MTTKRP's tile
        # columns and M / 16ceil(M/16) of the S build's tile rows
        for name, useful in (("xcov", M / (16.0 * ((M + 15) // 16))), ("mttkrp", R / (16.0 * ((R + 15) // 16)))):
            mfma[name]["useful_tile_fraction"] = useful
            mfma[name]["useful_TFLOPs"] = mfma[name]["TFLOPs"] * useful
            mfma[name]["useful_flop_utilisation"] = mfma[name]["mfma_utilisation"] * useful
        mfma["mttkrp"]["note"] = "TFLOPs / mfma_utilisation count the 16-wide MFMA tile the R=10 components occupy; useful_* scale by R/16"

    north = None
    if world == 1 and not args.no_north_star and (I_total, J, K, M) == (65536, 128, 128, 16):
        del X, Y
        if not args.no_fit:
            del st, sx, sm, X2, S, Mo
        try:
            north = north_star_leg(be, NipalsEngine, device, args.north_star_steps, min(args.repeats, 3), timer)
        except Exception as e:                     # never lose the headline line to the extra leg
            north = {"error": repr(e)}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu = cpu_baseline(J, K, M, R, args.noise, args.cpu_rows, I_total)

    cfg_name = {(65536, 128, 128, 16): "BASELINE configs[1]", (262144, 256, 256, 32): "BASELINE configs[4]"}.get(
        (I_total, J, K, M), "custom shape")
    if rank == 0:
        out = {
            "metric": "nipals_iters_per_sec", "value": args.steps / elapsed, "unit": "it/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "repeats": len(windows), "value_min": args.steps / max(windows), "value_max": args.steps / min(windows),
            "value_is": f"median of {len(windows)} timed windows of exactly {args.steps} iterations each",
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{cfg_name}: tPLS direct NIPALS iteration, X {I_total}x{J}x{K} f32 "
                                   f"(f64 accumulation), Y {I_total}x{M}, R={R}, noise {args.noise}",
                       "rows_per_gpu": rows, "parallelism": f"sample-mode shard x{world}" if world > 1 else "single GPU",
                       "rccl_ranks": dist.get_world_size() if backend == "nccl" else 0,
                       "x_reads_per_step": 2, "hip_graphs": bool(graphs_used), "graph_error": graph_error,
                       "eager_ms_per_step": eager_elapsed / args.steps * 1e3,
                       "eager_ms_per_step_min_max": [min(eager_windows) / args.steps * 1e3, max(eager_windows) / args.steps * 1e3]},
            "roofline": roofline, "cpu_baseline": cpu, "kernels": kern, "collectives": comm_info, "per_rank": per_rank,
            "mfma": mfma, "fit": fit_info, "north_star": north,
        }
        print(json.dumps(out), flush=True)
    if backend:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
