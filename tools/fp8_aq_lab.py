#!/usr/bin/env python3
"""Kernel-only timing of the quantise-on-load FP8 GEMM against the plain resident kernel at the o_proj / down shapes:
32 distinct strip-major weights per shape (the Infinity Cache cannot serve them), one HIP graph of 32 launches, replays
bracketed by events.  usage: python tools/fp8_aq_lab.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aphrodite_engine_amd import _custom_ops as ops  # noqa: E402
dev = "cuda"
M = 32
NL = 32


def timed(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s), torch.cuda.graph(g, stream=s):
        fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        g.replay()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) * 1e3 / (iters * NL)


for name, N, K, np_ in (("o_proj", 4096, 4096, 8), ("down_proj", 4096, 14336, 256)):
    gen = torch.Generator(device=dev).manual_seed(1)
    strips = [ops.fp8_strip_relayout((torch.randn(N, K, device=dev, generator=gen) * 0.5).to(torch.float8_e4m3fn), M) for _ in range(NL)]
    x = torch.randn(M, K, device=dev, generator=gen).half()
    part = x.float().abs().view(M, np_, -1).amax(2).contiguous()
    xp = torch.zeros(ops.aq_pairs_numel(M, K), dtype=x.dtype, device=dev)
    xp[ops.aq_pairs_index(M, K, dev).flatten()] = x.flatten()
    q, s = ops.scaled_fp8_quant(x, None, use_per_token_if_dynamic=True)
    t0 = timed(lambda: [ops.fp8_gemm_resident(q, st, slabs=True) for st in strips])
    t1 = timed(lambda: [ops.fp8_gemm_resident_aq(x, part, st) for st in strips])
    t2 = timed(lambda: [ops.fp8_gemm_resident_aq(xp, part, st, a_pairs=True) for st in strips])
    print(f"{name}: plain fp8 A {t0:.2f} us | AQ row-major {t1:.2f} | AQ pair-major {t2:.2f}   (APHRO_F8R_LAB={os.environ.get('APHRO_F8R_LAB', '0')})")
