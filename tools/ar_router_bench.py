#!/usr/bin/env python3
"""Loopback timing (one GPU, the real kernels, zero link time) of the sparse-MLP norm in front of the router under TP:
all-reduce + (norm + router) as two launches against the one-launch ROUTER form of the fused all-reduce + norm kernel.
usage: python tools/ar_router_bench.py [world] [tokens] [hidden] [experts]"""
import sys
import torch
sys.path.insert(0, ".")
from aphrodite_engine_amd import _custom_ops as ops
from aphrodite_engine_amd.distributed.custom_all_reduce import LoopbackAllreduce

world, tokens, hidden, E = (int(a) for a in (sys.argv[1:5] + ["4", "32", "4096", "8"][len(sys.argv) - 1:]))
dev = torch.device("cuda:0")
ca = LoopbackAllreduce(world, dev)
x = torch.randn(tokens, hidden, device=dev, dtype=torch.float16)
res = torch.randn(tokens, hidden, device=dev, dtype=torch.float16)
w = torch.rand(hidden, device=dev, dtype=torch.float16) + 0.5
gate = torch.randn(E, hidden, device=dev, dtype=torch.float16) * 0.05


def timed(fn, n=200):
    g = torch.cuda.CUDAGraph()
    fn()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (5 * n)


def two():
    o = ca.custom_all_reduce(x)
    ops.fused_add_rms_norm_router(o, None, res, True, w, 1e-5, gate)


def one():
    ca.fused_add_rms_norm_router(x, res, True, w, 1e-5, gate)


print(f"world {world} [{tokens}, {hidden}] E={E}: all-reduce + norm_router {timed(two):.2f} us   fused {timed(one):.2f} us   "
      f"all-reduce alone {timed(lambda: ca.custom_all_reduce(x)):.2f}   norm_router alone "
      f"{timed(lambda: ops.fused_add_rms_norm_router(x, None, res, True, w, 1e-5, gate)):.2f}")
ca.close()
