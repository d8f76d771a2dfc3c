#!/usr/bin/env python3
"""Is the decode attention kernel faster when its K/V blocks are already in the 256 MiB Infinity Cache?  (The upper bound of
any K/V prefetcher.)  configs[1] geometry, fused form; `ncache` caches cycled inside one HIP graph: 1 = the same 143 MB
every launch (warm in the Infinity Cache), 6 = 860 MB (cold)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aphrodite_engine_amd import _custom_ops as ops  # noqa: E402
from tools.prefetch_lab import timeit  # noqa: E402

B, H, HKV, HD, BS, ctx = 32, 32, 8, 128, 16, 1040
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
nblk_seq = (ctx + BS - 1) // BS
nblocks = B * nblk_seq
caches = [((torch.randn(nblocks, HKV, HD // 8, BS, 8, device=dev, generator=g) * 0.5).half(),
           (torch.randn(nblocks, HKV, HD, BS, device=dev, generator=g) * 0.5).half()) for _ in range(6)]
bt = torch.randperm(nblocks, device=dev, generator=g).to(torch.int32).view(B, nblk_seq).contiguous()
q = (torch.randn(B, H, HD, device=dev, generator=g) * 0.5).half()
slabs = torch.randn(2, B, (H + 2 * HKV) * HD, device=dev, generator=g) * 0.3
cos_sin = torch.randn(B, HD, device=dev, generator=g).half()
seq_lens = torch.full((B, ), ctx, dtype=torch.int32, device=dev)
slot = bt[:, (ctx - 1) // BS].long() * BS + (ctx - 1) % BS
nbytes = B * ctx * HKV * HD * 2 * 2
for form in ("plain", "fused"):
    for ncache in (6, 1, 2):
        def fn():
            for i in range(6):
                kc, vc = caches[i % ncache]
                if form == "plain":
                    o = torch.empty_like(q)
                    ops.paged_attention_v1(o, q, kc, vc, HKV, HD ** -0.5, bt, seq_lens, BS, ctx, None, "auto", 1.0, 1.0)
                else:
                    ops.paged_attention_rope_packed(slabs, None, cos_sin, slot, kc, vc, H, HKV, HD ** -0.5, bt, seq_lens, BS, ctx,
                                                    None, "auto", 1.0, 1.0)
        tt = timeit(fn, 6)
        print(json.dumps(dict(form=form, caches_cycled=ncache, MB=round(ncache * nbytes / 1e6), us=round(tt * 1e6, 2),
                              TBps=round(nbytes / tt / 1e12, 3))), flush=True)
