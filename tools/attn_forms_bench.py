#!/usr/bin/env python3
"""Decode attention at the configs[1] geometry (bs 32, 32 q / 8 kv heads, hd 128, block 16), us per launch inside a HIP graph with
the KV cold (caches cycled), for the forms the decode steps launch: plain (q given), fused (int4 step: qkv slabs + rotary + cache
write), scaled (FP8 step: the slabs of a W8A8 qkv GEMM dequantised on the fly), scaled + absmax partials + pair-major output."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aphrodite_engine_amd import _custom_ops as ops  # noqa: E402
dev = "cuda"
B, H, HKV, HD, BS = 32, 32, 8, 128, 16
ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 1040
g = torch.Generator(device=dev).manual_seed(0)
nblk_seq = (ctx + BS - 1) // BS
nblocks = B * nblk_seq
NC = 6
for kv in ("auto", "fp8"):
    esz = 1 if kv != "auto" else 2
    x = 16 // esz
    caches = []
    for _ in range(NC):
        if kv == "auto":
            caches.append(((torch.randn(nblocks, HKV, HD // x, BS, x, device=dev, generator=g) * 0.5).half(),
                           (torch.randn(nblocks, HKV, HD, BS, device=dev, generator=g) * 0.5).half()))
        else:
            caches.append((torch.randint(0, 120, (nblocks, HKV, HD // x, BS, x), device=dev, generator=g, dtype=torch.int32).to(torch.uint8),
                           torch.randint(0, 120, (nblocks, HKV, HD, BS), device=dev, generator=g, dtype=torch.int32).to(torch.uint8)))
    bt = torch.randperm(nblocks, device=dev, generator=g).to(torch.int32).view(B, nblk_seq).contiguous()
    seq_lens = torch.full((B, ), ctx, dtype=torch.int32, device=dev)
    q = (torch.randn(B, H, HD, device=dev, generator=g) * 0.5).half()
    ntot = (H + 2 * HKV) * HD
    slabs = [torch.randn(2, B, ntot, device=dev, generator=g) * 0.3 for _ in range(NC)]
    cos_sin = torch.randn(B, HD, device=dev, generator=g).half()
    slot = (bt[:, -1].long() * BS + (ctx - 1) % BS)
    row_sc = torch.rand(B, 1, device=dev, generator=g) + 0.5
    col_sc = torch.rand(ntot, device=dev, generator=g) + 0.5
    scale = HD ** -0.5
    for form in ("plain", "fused", "scaled", "scaled_absmax"):
        def launches():
            for i, (kc, vc) in enumerate(caches):
                if form == "plain":
                    o = torch.empty_like(q)
                    ops.paged_attention_v1(o, q, kc, vc, HKV, scale, bt, seq_lens, BS, ctx, None, kv, 1.0, 1.0)
                elif form == "fused":
                    ops.paged_attention_rope_packed(slabs[i], None, cos_sin, slot, kc, vc, H, HKV, scale, bt, seq_lens, BS, ctx, None, kv, 1.0, 1.0)
                else:
                    ops.paged_attention_rope_scaled(slabs[i], row_sc, col_sc, None, cos_sin, slot, kc, vc, H, HKV, scale, bt, seq_lens, BS, ctx,
                                                    None, kv, 1.0, 1.0, want_absmax=form == "scaled_absmax", out_pairs=form == "scaled_absmax")
        launches()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            launches()
        for _ in range(3):
            gr.replay()
        torch.cuda.synchronize()
        s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_.record()
        for _ in range(10):
            gr.replay()
        e_.record()
        e_.synchronize()
        print(f"kv {kv:5s} {form:14s} ctx {ctx}: {s_.elapsed_time(e_) * 1e3 / (10 * NC):6.2f} us")
