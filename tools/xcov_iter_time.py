#!/usr/bin/env python3
"""Where does an inner iteration of the cross-covariance loop spend its time?  One iteration on S (M x P, here 16 x 16384) is
~15 dependent launches.  Prints, per iteration: the host time to ENQUEUE it (cmtfpls_xcov_iterate_f64, no wait), the time per
iteration of a long back-to-back chain (eager), and the same chain replayed from a HIP graph.
Usage: python tools/xcov_iter_time.py [n_squarings]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmtf_pls_amd.backend import HipBackend  # noqa: E402

nsq = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
be = HipBackend(dev)
M, A, B = 16, 128, 128
g = torch.Generator(device="cpu").manual_seed(3)
S = (torch.randn(M, 4, generator=g, dtype=torch.float64) @ torch.randn(4, A * B, generator=g, dtype=torch.float64)
     + 0.3 * torch.randn(M, A * B, generator=g, dtype=torch.float64)).to(dev)
G = torch.eye(M, dtype=torch.float64, device=dev)
q = [be.zeros(M), be.zeros(M)]
q[0][0] = 1.0
Z, wA, wB = be.empty(A * B), be.empty(A), be.empty(B)
status = be.zeros(3)


def one(i):
    be.xcov_iterate(S, A, B, q[i & 1], Z, wA, wB, status[1:3], nsq, q[(i + 1) & 1], G, status[0:1], True)


N = 400
for i in range(20):
    one(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(N):
    one(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"eager, {nsq} squarings: enqueue {1e6 * (t1 - t0) / N:.1f} us/iteration on the host, {1e6 * (t2 - t0) / N:.1f} us/iteration end to end", flush=True)

graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    one(0)
    one(1)
torch.cuda.synchronize()
for _ in range(5):
    graph.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N // 2):
    graph.replay()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"graph of 2 iterations: enqueue {1e6 * (t1 - t0) / N:.1f} us/iteration on the host, {1e6 * (t2 - t0) / N:.1f} us/iteration end to end", flush=True)
# the waiting loop: one status copy + synchronisation per iteration
host = torch.empty(3, dtype=torch.float64, pin_memory=True)
t0 = time.perf_counter()
for i in range(N):
    one(i)
    host.copy_(status, non_blocking=True)
    torch.cuda.current_stream().synchronize()
t2 = time.perf_counter()
print(f"eager + status copy + synchronise every iteration: {1e6 * (t2 - t0) / N:.1f} us/iteration", flush=True)
# the pipelined loop: iteration i + 1 enqueued (pre-marshalled arguments) before the host waits for the status of iteration i
qq = [be.zeros(M), be.zeros(M), be.zeros(M)]
qq[0][0] = 1.0
sets = [(be.empty(A * B), be.empty(A), be.empty(B), be.zeros(3)) for _ in range(2)]
plans = [be.xcov_iterate_plan(S, A, B, qq[i % 3], sets[i & 1][0], sets[i & 1][1], sets[i & 1][2], sets[i & 1][3], qq[(i + 1) % 3], G) for i in range(6)]
for i in range(12):
    plans[i % 6](nsq, True)
torch.cuda.synchronize()
t0 = time.perf_counter()
plans[0](nsq, True)
tok = be.status_snapshot(sets[0][3], 0)
t_enq = t_wait = 0.0
for i in range(N):
    ta = time.perf_counter()
    plans[(i + 1) % 6](nsq, True)
    nxt = be.status_snapshot(sets[(i + 1) & 1][3], (i + 1) & 1)
    tb = time.perf_counter()
    host = be.status_wait(tok)
    tc = time.perf_counter()
    t_enq += tb - ta
    t_wait += tc - tb
    tok = nxt
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"pipelined (plans + status_to_host): {1e6 * (t2 - t0) / N:.1f} us/iteration; host: enqueue {1e6 * t_enq / N:.1f} us, waiting {1e6 * t_wait / N:.1f} us", flush=True)
