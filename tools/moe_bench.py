#!/usr/bin/env python3
"""configs[4] MoE layer timing: Mixtral-8x7B GPTQ-int4 experts (TP = --tp slices), decode batch,
fused path (routing + grouped GEMMs, aphrodite_engine_amd/moe.py) vs the reference's dense
per-expert loop (mixtral_quant.py:130-156) over the same device GEMM ops.  HIP-graph replay timing;
several weight copies so the Infinity Cache cannot serve the experts."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aphrodite_engine_amd import _custom_ops as ops  # noqa: E402
from aphrodite_engine_amd import moe as M  # noqa: E402


def rand_gptq(k, n, dev, g):
    qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (k // 8, n), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
    qz = torch.randint(-2 ** 31, 2 ** 31 - 1, (k // 128, n // 8), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
    sc = (torch.rand(k // 128, n, generator=g, device=dev) * 0.01).half()
    return qw, qz, sc


def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        fn()
    gr.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        gr.replay()
    e.record(); e.synchronize()
    return s.elapsed_time(e) * 1e-3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tp", type=int, default=4)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--layers", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(0)
    E, topk, H, I = 8, 2, 4096, 14336 // a.tp
    layers = []
    for _ in range(a.layers):
        w13 = [rand_gptq(H, 2 * I, dev, g) for _ in range(E)]
        w2 = [rand_gptq(I, H, dev, g) for _ in range(E)]
        layers.append((M.Wna16Experts(w13, w2), w13, w2))
    x = torch.randn(a.batch, H, device=dev, dtype=torch.float16) * 0.5
    gating = torch.randn(a.batch, E, device=dev)
    empty = torch.empty(0, dtype=torch.int32, device=dev)

    def fused():
        for ex, _, _ in layers:
            M.fused_wna16_moe(x, ex, gating, topk, True)

    shuf = [([ops.gptq_marlin_repack(q, empty, H, 2 * I, 4) for q, _, _ in w13],
             [ops.gptq_marlin_repack(q, empty, I, H, 4) for q, _, _ in w2]) for _, w13, w2 in layers]

    def dense():
        for li, (_, w13, w2) in enumerate(layers):
            rw, rids = M.fused_topk(x, gating, topk, True)
            out = None
            for e in range(E):
                h = ops.gptq_gemm(x, shuf[li][0][e], w13[e][1], w13[e][2], empty, True, 4)
                act = torch.empty(a.batch, I, dtype=torch.float16, device=dev)
                ops.silu_and_mul(act, h)
                y = ops.gptq_gemm(act, shuf[li][1][e], w2[e][1], w2[e][2], empty, True, 4)
                y = y * (rw * (rids == e)).sum(dim=-1, keepdim=True).half()
                out = y if out is None else out + y

    tf = timeit(fused) / a.layers
    td = timeit(dense) / a.layers
    wbytes_all = E * (H * 2 * I + I * H) // 2 * (1 + 4.5 / 64)
    sel = len(np.unique(M.fused_topk(x, gating, topk, True)[1].cpu().numpy()))
    print(json.dumps({"config": f"Mixtral-8x7B GPTQ g128 experts, TP{a.tp} slice, batch {a.batch}, top-2 of 8",
                      "fused_us": round(tf * 1e6, 1), "dense_loop_us": round(td * 1e6, 1),
                      "speedup": round(td / tf, 2), "active_experts": sel,
                      "expert_weight_bytes_read_MB": round(wbytes_all * sel / E / 1e6, 1),
                      "fused_GBps": round(wbytes_all * sel / E / tf / 1e9, 1)}))


if __name__ == "__main__":
    main()
