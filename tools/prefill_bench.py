#!/usr/bin/env python3
"""Prefill-side timing (configs[2]: seq 8192): causal varlen flash attention and the large-M W4A16 /
FP8 linear path, reported against the dense MFMA peak (2.5 PFLOP/s f16/bf16, MI355X_MICROARCH.md)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aphrodite_engine_amd import _custom_ops as ops  # noqa: E402


def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) * 1e-3 / iters


def main():
    dev = "cuda"
    out = []
    for T in (2048, 8192):
        Hq, Hkv, D = 32, 8, 128
        qkv = torch.randn(T, (Hq + 2 * Hkv) * D, device=dev, dtype=torch.float16) * 0.5
        q = qkv[:, :Hq * D].view(T, Hq, D)
        k = qkv[:, Hq * D:(Hq + Hkv) * D].view(T, Hkv, D)
        v = qkv[:, (Hq + Hkv) * D:].view(T, Hkv, D)
        cu = torch.tensor([0, T], dtype=torch.int32, device=dev)
        t = timeit(lambda: ops.flash_attn_varlen(q, k, v, cu, T, D ** -0.5, causal=True))
        flops = 4 * T * T * D * Hq / 2
        out.append({"kernel": "flash_attn_varlen causal", "T": T, "ms": round(t * 1e3, 3),
                    "TFLOPs": round(flops / t / 1e12, 1), "frac_of_2.5PF": round(flops / t / 2.5e15, 3)})
    # large-M W4A16 (the hand-written MFMA kernel; round 1: dequant + library GEMM)
    K, N, M = 4096, 28672, 8192
    g = torch.Generator(device=dev).manual_seed(0)
    qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // 8, N), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
    qz = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // 128, N // 8), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
    sc = (torch.rand(K // 128, N, generator=g, device=dev) * 0.01).half()
    a = torch.randn(M, K, device=dev, dtype=torch.float16)
    empty = torch.empty(0, dtype=torch.int32, device=dev)
    t = timeit(lambda: ops.gptq_gemm(a, qw, qz, sc, empty, True, 4), 3)
    flops = 2.0 * M * K * N
    out.append({"kernel": "gptq_gemm M=8192 (wna16_gemm_large_kernel)", "ms": round(t * 1e3, 3),
                "TFLOPs": round(flops / t / 1e12, 1), "frac_of_2.5PF": round(flops / t / 2.5e15, 3)})
    t = timeit(lambda: ops.gptq_dequant(qw, qz, sc, empty, True, 4), 3)
    out.append({"kernel": "gptq_dequant 4096x28672", "ms": round(t * 1e3, 3),
                "GBps": round((K * N / 2 + K * N * 2) / t / 1e9, 1)})
    for o in out:
        print(json.dumps(o))


if __name__ == "__main__":
    main()


def context_bench():
    """Chunked prefill: 2048 new tokens on top of 6144 cached ones (paged fp16 cache), Llama-3-8B heads."""
    import os
    dev = "cuda"
    Hq, Hkv, D, BS, ctx, new = 32, 8, 128, 16, 6144, 2048
    nblk = (ctx + new + BS - 1) // BS
    kc = (torch.randn(nblk, Hkv, D // 8, BS, 8, device=dev) * 0.5).half()
    vc = (torch.randn(nblk, Hkv, D, BS, device=dev) * 0.5).half()
    bt = torch.randperm(nblk, device=dev).to(torch.int32).view(1, nblk)
    qkv = torch.randn(new, (Hq + 2 * Hkv) * D, device=dev, dtype=torch.float16) * 0.5
    q = qkv[:, :Hq * D].view(new, Hq, D); k = qkv[:, Hq * D:(Hq + Hkv) * D].view(new, Hkv, D); v = qkv[:, (Hq + Hkv) * D:].view(new, Hkv, D)
    out = torch.empty(new, Hq, D, device=dev, dtype=torch.float16)
    i32 = lambda *a: torch.tensor(a, dtype=torch.int32, device=dev)
    args = ("auto", kc, vc, bt, i32(0, new), i32(ctx + new), i32(ctx), new, 1.0, 1.0, None, None)
    flops = 4.0 * D * Hq * (new * ctx + new * new / 2)
    for label, env in (("gathered context + v3 kernel", None), ("per-element cache gather kernel", "1")):
        if env:
            os.environ["APHRO_CA_NO_GATHER"] = env
        t = timeit(lambda: ops.context_attention_fwd(q, k, v, out, *args, max_seq_len=ctx + new, total_kv_tokens=ctx + new))
        os.environ.pop("APHRO_CA_NO_GATHER", None)
        print(json.dumps({"kernel": f"context_attention_fwd 6144 cached + 2048 new ({label})", "ms": round(t * 1e3, 3),
                          "TFLOPs": round(flops / t / 1e12, 1), "frac_of_2.5PF": round(flops / t / 2.5e15, 3)}))


if __name__ == "__main__" and os.environ.get("PREFILL_BENCH_CONTEXT", "1") == "1":
    context_bench()
