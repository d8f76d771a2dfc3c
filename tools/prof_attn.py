#!/usr/bin/env python3
"""Runs the decode attention kernel a few times (for rocprofv3 --pmc / --kernel-trace)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aphrodite_engine_amd import _custom_ops as ops  # noqa: E402
ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 1040
kvd = sys.argv[2] if len(sys.argv) > 2 else "auto"
bs, Hq, Hkv, D, BS = 32, 32, 8, 128, 16
bps = (ctx + BS - 1) // BS
nb = bs * bps
caches = []
for _ in range(6):
    if kvd == "auto":
        caches.append((torch.randn(nb, Hkv, D // 8, BS, 8, device="cuda", dtype=torch.float16) * 0.1,
                       torch.randn(nb, Hkv, D, BS, device="cuda", dtype=torch.float16) * 0.1))
    else:
        caches.append((torch.randint(0, 0x48, (nb, Hkv, D // 16, BS, 16), device="cuda", dtype=torch.uint8),
                       torch.randint(0, 0x48, (nb, Hkv, D, BS), device="cuda", dtype=torch.uint8)))
bt = torch.randperm(nb, device="cuda").view(bs, bps).int()
sl = torch.full((bs, ), ctx, dtype=torch.int32, device="cuda")
q = torch.randn(bs, Hq, D, device="cuda", dtype=torch.float16)
o = torch.empty_like(q)
mode = sys.argv[3] if len(sys.argv) > 3 else "v1"
slabs = torch.randn(2, bs, (Hq + 2 * Hkv) * D, device="cuda") * 0.3
pos = torch.full((bs, ), ctx - 1, dtype=torch.int64, device="cuda")
cos_sin = torch.randn(ctx + 8, D, device="cuda").half()
slots = (bt[:, (ctx - 1) // BS].long() * BS + (ctx - 1) % BS)
for _ in range(5):
    for kc, vc in caches:
        if mode == "v1":
            ops.paged_attention_v1(o, q, kc, vc, Hkv, D ** -0.5, bt, sl, BS, ctx, None, kvd, 1.0, 1.0)
        elif mode == "packed":
            ops.paged_attention_packed(q, kc, vc, Hkv, D ** -0.5, bt, sl, BS, ctx, None, kvd, 1.0, 1.0)
        else:
            ops.paged_attention_rope_packed(slabs, pos, cos_sin, slots, kc, vc, Hq, Hkv, D ** -0.5, bt, sl, BS,
                                            ctx, None, kvd, 1.0, 1.0)
torch.cuda.synchronize()
