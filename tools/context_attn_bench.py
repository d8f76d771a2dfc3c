#!/usr/bin/env python3
"""Round 3: prefill with cached context outside the third-generation kernel's sweet spot (head 128, >= 1024 keys, no
window): gather-once + second / first generation tile machines (the default) vs the round-1 scalar-gather kernel
(APHRO_CA_NO_GATHER=1).  HIP events over 10 calls."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aphrodite_engine_amd import _custom_ops as ops  # noqa: E402

CASES = [  # name, batch, ctx, new, Hq, Hkv, D, window
    ("hd128 short: 8 x (512 cached + 256 new)", 8, 512, 256, 32, 8, 128, None),
    ("hd128 window 1024: 4096 cached + 1024 new", 1, 4096, 1024, 32, 8, 128, 1024),
    ("hd64: 4 x (2048 cached + 512 new)", 4, 2048, 512, 32, 8, 64, None),
    ("hd96: 4 x (1024 cached + 512 new)", 4, 1024, 512, 24, 8, 96, None),
    ("hd256: 2 x (1024 cached + 512 new)", 2, 1024, 512, 8, 4, 256, None),
    ("hd128 long (third generation): 6144 cached + 2048 new", 1, 6144, 2048, 32, 8, 128, None),
]


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) * 1e-3 / iters


out = []
g = torch.Generator(device="cuda").manual_seed(0)
for name, B, ctx, new, Hq, Hkv, D, win in CASES:
    BS = 16
    nblk_seq = (ctx + new + BS - 1) // BS
    nb = B * nblk_seq
    kc = (torch.randn(nb, Hkv, D // 8, BS, 8, device="cuda", generator=g) * 0.5).half()
    vc = (torch.randn(nb, Hkv, D, BS, device="cuda", generator=g) * 0.5).half()
    bt = torch.randperm(nb, device="cuda", generator=g).to(torch.int32).view(B, nblk_seq)
    T = B * new
    qkv = torch.randn(T, (Hq + 2 * Hkv) * D, device="cuda", dtype=torch.float16, generator=g) * 0.5
    q, k, v = qkv[:, :Hq * D].view(T, Hq, D), qkv[:, Hq * D:(Hq + Hkv) * D].view(T, Hkv, D), qkv[:, (Hq + Hkv) * D:].view(T, Hkv, D)
    o = torch.empty(T, Hq, D, device="cuda", dtype=torch.float16)
    i32 = lambda x: torch.tensor(x, dtype=torch.int32, device="cuda")
    start = i32([i * new for i in range(B + 1)])
    args = ("auto", kc, vc, bt, start, i32([ctx + new] * B), i32([ctx] * B), new, 1.0, 1.0, None, win)
    kw = dict(max_seq_len=ctx + new, total_kv_tokens=B * (ctx + new))
    keys = (ctx + new / 2) if win is None else min(win, ctx + new / 2)
    fl = 4.0 * D * Hq * B * new * keys
    rec = {"case": name}
    for tag, env in (("tile machines", None), ("scalar-gather kernel (round 1)", "1")):
        if env:
            os.environ["APHRO_CA_NO_GATHER"] = env
        else:
            os.environ.pop("APHRO_CA_NO_GATHER", None)
        t = timeit(lambda: ops.context_attention_fwd(q, k, v, o, *args, **kw))
        rec[tag] = {"ms": round(t * 1e3, 3), "TFLOPs": round(fl / t / 1e12, 1)}
    os.environ.pop("APHRO_CA_NO_GATHER", None)
    out.append(rec)
    print(json.dumps(rec), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "context_attn_bench.json"), "w"), indent=1)
