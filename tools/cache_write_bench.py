#!/usr/bin/env python3
"""reshape_and_cache on an 8192-token Llama-3-8B prompt (8 KV heads x 128, block 16, shuffled block table), us per call:
the 16-token window form (>= 64 tokens) against the per-token form (the same tokens in calls of 63)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aphrodite_engine_amd import _custom_ops as ops  # noqa: E402

DEV = "cuda:0"
T, H, D, BS = 8192, 8, 128, 16
for kv in ("auto", "fp8"):
    cdt = torch.float16 if kv == "auto" else torch.uint8
    x = 8 if kv == "auto" else 16
    nb = T // BS
    kc = torch.zeros(nb, H, D // x, BS, x, dtype=cdt, device=DEV)
    vc = torch.zeros(nb, H, D, BS, dtype=cdt, device=DEV)
    qkv = torch.randn(T, (32 + 2 * H) * D, device=DEV, dtype=torch.float16)
    k = qkv[:, 32 * D:(32 + H) * D].view(T, H, D)
    v = qkv[:, (32 + H) * D:].view(T, H, D)
    bt = torch.randperm(nb, device=DEV)
    pos = torch.arange(T, device=DEV)
    slots = bt[pos // BS] * BS + pos % BS

    def timed(fn, n=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    t_new = timed(lambda: ops.reshape_and_cache(k, v, kc, vc, slots, kv, 0.5, 0.5))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for c0 in range(0, T, 63):
            ops.reshape_and_cache(k[c0:c0 + 63], v[c0:c0 + 63], kc, vc, slots[c0:c0 + 63], kv, 0.5, 0.5)
    t_old = timed(g.replay)
    mb = T * H * D * 2 * (2 + (2 if kv == "auto" else 1)) / 1e6
    print(f"kv_cache={kv}: window form {t_new:.1f} us ({mb / t_new:.2f} TB/s of K/V read + cache written)   the per-token form, same tokens in 131 calls of 63 "
          f"(graph; launch-bound -- the one-call figure of the per-token form is the trace's 55.3 us, profiles/r6_prefill_e2e_trace.txt) {t_old:.1f} us")
