"""all-reduce -> fused_add_rms_norm_pack as two launches vs the one-launch form (csrc/custom_all_reduce.hip), on a LOOPBACK
communicator (one GPU, every peer = this rank's own memory: the kernels' real instruction stream without link time).
32 dependent repetitions captured into one HIP graph, as the decode step runs them.
    python tools/ar_norm_bench.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aphrodite_engine_amd import _custom_ops as ops  # noqa: E402
from aphrodite_engine_amd.distributed.custom_all_reduce import LoopbackAllreduce  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, n=32, iters=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters / n * 1e6


for world, tokens, hidden in ((8, 64, 8192), (8, 32, 8192), (4, 32, 4096), (2, 32, 4096), (8, 64, 4096)):
    ca = LoopbackAllreduce(world, dev)
    x = torch.randn(tokens, hidden, device=dev, dtype=torch.float16)
    res = torch.randn(tokens, hidden, device=dev, dtype=torch.float16)
    w = torch.ones(hidden, device=dev, dtype=torch.float16)
    pf = torch.empty(32 * 1024 * 1024, dtype=torch.uint8, device=dev)
    one_shot = ops.custom_ar_fused_norm_one_shot(world, tokens, hidden, 2)

    def two():
        s = ca.custom_all_reduce(x)
        ops.fused_add_rms_norm_pack(s, None, res, True, w, 1e-5)

    def ar_only():
        ca.custom_all_reduce(x)

    def norm_only():
        ops.fused_add_rms_norm_pack(x, None, res, True, w, 1e-5)

    def one(prefetch=None):
        return lambda: ca.fused_add_rms_norm(x, res, True, w, 1e-5, prefetch=prefetch)
    print(f"world {world} [{tokens}, {hidden}] ({'one-shot' if one_shot else 'two-shot'}): all-reduce {timed(ar_only):6.2f} us | norm {timed(norm_only):6.2f} | "
          f"two launches {timed(two):6.2f} | one launch {timed(one()):6.2f} | one launch + 32 MiB weight prefetch {timed(one(pf)):6.2f}", flush=True)
    ca.close()
