#!/usr/bin/env python3
"""Round 3: decode attention with split-KV inside the launch (APHRO_PA_SPLITS=n) at the bench geometries: configs[1]
(bs 32, 8 kv heads: 256 groups), the configs[3] TP8 shard (bs 64, ONE kv head: 64 groups) and the configs[4] TP4 shard
(bs 32, 2 kv heads).  KV cold (cycling over distinct caches), HIP-graph replay."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aphrodite_engine_amd import _custom_ops as ops  # noqa: E402
from tools.prefetch_lab import timeit  # noqa: E402

OUT = []
for name, bs, Hq, Hkv, ctx in (("cfg1 bs32 8kvh", 32, 32, 8, 1040), ("cfg3 tp8 shard bs64 1kvh", 64, 8, 1, 1040),
                               ("cfg4 tp4 shard bs32 2kvh", 32, 8, 2, 1040), ("cfg1 ctx4096", 32, 32, 8, 4096)):
    D, BS = 128, 16
    bps = (ctx + BS - 1) // BS
    nb = bs * bps
    per = 2 * bs * ctx * Hkv * D * 2
    ncache = max(2, min(12, (700 << 20) // per))
    caches = [(torch.randn(nb, Hkv, D // 8, BS, 8, device="cuda", dtype=torch.float16) * 0.1,
               torch.randn(nb, Hkv, D, BS, device="cuda", dtype=torch.float16) * 0.1) for _ in range(ncache)]
    bt = torch.randperm(nb, device="cuda").view(bs, bps).int()
    sl = torch.full((bs, ), ctx, dtype=torch.int32, device="cuda")
    q = torch.randn(bs, Hq, D, device="cuda", dtype=torch.float16)
    ntot = (Hq + 2 * Hkv) * D
    slabs = torch.randn(2, bs, ntot, device="cuda") * 0.1
    cs = torch.randn(bs, D, device="cuda").half()
    slots = (bt[:, (ctx - 1) // BS].long() * BS + (ctx - 1) % BS)
    L = max(12, ncache)
    for sp in (1, 2, 3, 4, 8, 0):
        if sp:
            os.environ["APHRO_PA_SPLITS"] = str(sp)
        else:
            os.environ.pop("APHRO_PA_SPLITS", None)       # the library's own choice

        def plain():
            for i in range(L):
                kc, vc = caches[i % ncache]
                ops.paged_attention_packed(q, kc, vc, Hkv, D ** -0.5, bt, sl, BS, ctx, None, "auto", 1.0, 1.0)

        def fused():
            for i in range(L):
                kc, vc = caches[i % ncache]
                ops.paged_attention_rope_packed(slabs, None, cs, slots, kc, vc, Hq, Hkv, D ** -0.5, bt, sl, BS, ctx, None,
                                                "auto", 1.0, 1.0)
        for form, fn in (("plain", plain), ("rope-fused", fused)):
            tt = timeit(fn, L)
            rec = dict(case=name, form=form, splits=sp or "auto", us=round(tt * 1e6, 2), TBps=round(per / tt / 1e12, 3))
            OUT.append(rec)
            print(json.dumps(rec), flush=True)
    del caches
    torch.cuda.empty_cache()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "attn_split_bench.jsonl"), "w") as f:
    for r in OUT:
        f.write(json.dumps(r) + "\n")
