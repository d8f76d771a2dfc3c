"""Grouped FP8 W8A8 MoE GEMMs at Mixtral-8x7B decode shapes (one GPU, all 8 experts active): microseconds per launch by
HIP-graph replay, GB/s over the expert weights.  Knobs: APHRO_FP8_MOE_NW (K-split waves), APHRO_FP8_MOE_NT (n-tiles per
workgroup) -- read per call, so one process sweeps them.

    python tools/fp8_moe_bench.py [--tokens 32]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aphrodite_engine_amd import _custom_ops as ops  # noqa: E402


def graph_us(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=32)
    ap.add_argument("--experts", type=int, default=8)
    ap.add_argument("--topk", type=int, default=2)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    e, topk, t = args.experts, args.topk, args.tokens
    hidden, inter = 4096, 14336
    slots = t * topk
    # balanced routing: slot s -> expert s % e
    ids = (torch.arange(slots, device=dev, dtype=torch.int32) % e).reshape(t, topk).contiguous()
    max_pad = slots + e * 15
    sorted_ids = torch.empty(max_pad, dtype=torch.int32, device=dev)
    expert_ids = torch.empty((max_pad + 15) // 16, dtype=torch.int32, device=dev)
    post_pad = torch.empty(1, dtype=torch.int32, device=dev)
    ops.moe_align_block_size(ids, e, 16, sorted_ids, expert_ids, post_pad)
    one = torch.ones(1, dtype=torch.float32, device=dev)
    ws = torch.ones(e, dtype=torch.float32, device=dev)
    tw = torch.rand(slots, dtype=torch.float32, device=dev)
    for name, n, k, div, rows in (("w13", 2 * inter, hidden, topk, t), ("w2", hidden, inter, 1, slots)):
        w = torch.randint(0, 120, (e, n, k), dtype=torch.uint8, device=dev).view(torch.float8_e4m3fn)
        a = torch.randint(0, 120, (rows, k), dtype=torch.uint8, device=dev).view(torch.float8_e4m3fn)
        out = torch.empty(slots, n, dtype=torch.float16, device=dev)
        ref = None
        for nw in ("8", "4"):
            for nt in ("4", "2", "1"):
                os.environ["APHRO_FP8_MOE_NW"], os.environ["APHRO_FP8_MOE_NT"] = nw, nt
                fn = lambda: ops.fp8_moe_gemm(a, w, one, ws, tw if div == 1 else None, sorted_ids, expert_ids, post_pad, out, div)
                us = graph_us(fn)
                if ref is None:
                    ref = out.float().clone()
                err = (out.float() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-9)
                print(f"{name} [E={e}, N={n}, K={k}] slots={slots} NW={nw} NT={nt}: {us:8.1f} us  {w.numel() / us / 1e3:7.0f} GB/s"
                      f"  max|d|/max|ref| vs first = {err:.2e}", flush=True)


if __name__ == "__main__":
    main()
