#!/usr/bin/env python3
"""Decode attention (fused rotary + cache write form and plain form) at the configs[1] geometry over context lengths
around a multiple of the 32-token x 8-wave round: what the LAST, partially filled round of a workgroup costs.
KV cold (three caches cycled inside one HIP graph), HIP-event timing."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aphrodite_engine_amd import _custom_ops as ops  # noqa: E402
from tools.prefetch_lab import timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ctx", type=int, nargs="+", default=[992, 1024, 1025, 1040, 1056, 1057, 1120, 1152, 1280])
    ap.add_argument("--kv", nargs="+", default=["auto", "fp8"])
    ap.add_argument("--batch", type=int, default=32)
    args = ap.parse_args()
    B, H, HKV, HD, BS = args.batch, 32, 8, 128, 16
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    rows = []
    for kv in args.kv:
        esz = 1 if kv != "auto" else 2
        x = 16 // esz
        maxctx = max(args.ctx)
        nblk_seq = (maxctx + BS - 1) // BS
        nblocks = B * nblk_seq
        caches = []
        for _ in range(3):
            if kv == "auto":
                kc = (torch.randn(nblocks, HKV, HD // x, BS, x, device=dev, generator=g) * 0.5).half()
                vc = (torch.randn(nblocks, HKV, HD, BS, device=dev, generator=g) * 0.5).half()
            else:
                kc = torch.randint(0, 120, (nblocks, HKV, HD // x, BS, x), device=dev, generator=g, dtype=torch.int32).to(torch.uint8)
                vc = torch.randint(0, 120, (nblocks, HKV, HD, BS), device=dev, generator=g, dtype=torch.int32).to(torch.uint8)
            caches.append((kc, vc))
        bt = torch.randperm(nblocks, device=dev, generator=g).to(torch.int32).view(B, nblk_seq).contiguous()
        q = (torch.randn(B, H, HD, device=dev, generator=g) * 0.5).half()
        slabs = torch.randn(2, B, (H + 2 * HKV) * HD, device=dev, generator=g) * 0.3
        cos_sin = torch.randn(B, HD, device=dev, generator=g).half()
        scale = HD ** -0.5
        for ctx in args.ctx:
            seq_lens = torch.full((B, ), ctx, dtype=torch.int32, device=dev)
            slot = bt[:, (ctx - 1) // BS].long() * BS + (ctx - 1) % BS
            nbytes = B * ctx * HKV * HD * 2 * esz
            for form in ("plain", "fused"):
                def fn():
                    for kc, vc in caches:
                        if form == "plain":
                            o = torch.empty_like(q)
                            ops.paged_attention_v1(o, q, kc, vc, HKV, scale, bt, seq_lens, BS, ctx, None, kv, 1.0, 1.0)
                        else:
                            ops.paged_attention_rope_packed(slabs, None, cos_sin, slot, kc, vc, H, HKV, scale, bt, seq_lens, BS,
                                                            ctx, None, kv, 1.0, 1.0)
                tt = timeit(fn, len(caches))
                r = dict(kv=kv, form=form, ctx=ctx, pairs=(ctx + 31) // 32, us=round(tt * 1e6, 2), TBps=round(nbytes / tt / 1e12, 3))
                rows.append(r)
                print(json.dumps(r), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "attn_ctx_sweep.jsonl"), "w") as f:
        for r in rows:
            f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
