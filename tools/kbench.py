#!/usr/bin/env python3
"""Per-kernel microbenchmark of the hot kernels at the BASELINE configs[1]
shapes (Llama-3-8B, M = batch 32).  Each shape cycles over enough distinct
weight copies that the 256 MiB Infinity Cache cannot serve them; timing is by
HIP events on the launching stream.  Env knobs of the C library
(APHRO_WNA16_VEC / APHRO_WNA16_KSPLIT / ...) can be swept with --sweep."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aphrodite_engine_amd import _custom_ops as ops  # noqa: E402

SHAPES = {"qkv": (4096, 6144), "o": (4096, 4096), "gate_up": (4096, 28672), "down": (14336, 4096)}


def timeit(fn, n_launch, iters=10):
    """HIP-graph replay timing: no host launch overhead between the kernels."""
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        g.replay()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) * 1e-3 / (iters * n_launch)


def gptq_case(name, M, copies):
    K, N = SHAPES[name]
    G = K // 128
    g = torch.Generator(device="cuda").manual_seed(0)
    ws = []
    for _ in range(copies):
        qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // 8, N), generator=g, device="cuda", dtype=torch.int64).to(torch.int32)
        qz = torch.randint(-2 ** 31, 2 ** 31 - 1, (G, N // 8), generator=g, device="cuda", dtype=torch.int64).to(torch.int32)
        sc = (torch.rand(G, N, generator=g, device="cuda") * 0.01).half()
        ws.append((qw, qz, sc))
    a = torch.randn(M, K, device="cuda", dtype=torch.float16)
    empty = torch.empty(0, dtype=torch.int32, device="cuda")
    nbytes = K * N // 2 + G * N * 2 + G * N // 2 + M * K * 2 + M * N * 2

    def run():
        for qw, qz, sc in ws:
            ops.gptq_gemm(a, qw, qz, sc, empty, True, 4)
    return run, len(ws), nbytes


def fp8_case(name, M, copies):
    K, N = SHAPES[name]
    ws = [(torch.randn(N, K, device="cuda") * 10).to(torch.float8_e4m3fn) for _ in range(copies)]
    a = torch.randn(M, K, device="cuda").to(torch.float8_e4m3fn)
    sa = torch.rand(M, 1, device="cuda")
    sb = torch.rand(N, 1, device="cuda")
    nbytes = K * N + M * K + M * N * 2

    def run():
        for w in ws:
            ops.cutlass_scaled_mm(a, w.t(), sa, sb, torch.float16)
    return run, len(ws), nbytes


def attn_case(bs, ctx, kv_dtype, layers):
    Hq, Hkv, D, BS = 32, 8, 128, 16
    bps = (ctx + BS - 1) // BS
    nb = bs * bps
    cdt = torch.float16 if kv_dtype == "auto" else torch.uint8
    x = 8 if kv_dtype == "auto" else 16
    caches = []
    for _ in range(layers):
        if kv_dtype == "auto":
            kc = torch.randn(nb, Hkv, D // x, BS, x, device="cuda", dtype=cdt) * 0.1
            vc = torch.randn(nb, Hkv, D, BS, device="cuda", dtype=cdt) * 0.1
        else:
            kc = torch.randint(0, 0x48, (nb, Hkv, D // x, BS, x), device="cuda", dtype=torch.uint8)
            vc = torch.randint(0, 0x48, (nb, Hkv, D, BS), device="cuda", dtype=torch.uint8)
        caches.append((kc, vc))
    bt = torch.randperm(nb, device="cuda").view(bs, bps).int()
    sl = torch.full((bs, ), ctx, dtype=torch.int32, device="cuda")
    q = torch.randn(bs, Hq, D, device="cuda", dtype=torch.float16)
    o = torch.empty_like(q)
    P = (ctx + 511) // 512
    tmp = torch.empty(bs, Hq, P, D, device="cuda", dtype=torch.float16)
    es = torch.empty(bs, Hq, P, device="cuda")
    ml = torch.empty_like(es)
    esz = 2 if kv_dtype == "auto" else 1
    nbytes = 2 * bs * ctx * Hkv * D * esz + 2 * bs * Hq * D * 2

    def run_rocm():
        for kc, vc in caches:
            ops.paged_attention_rocm(o, es, ml, tmp, q, kc, vc, Hkv, D ** -0.5, bt, sl, BS, ctx, None,
                                     kv_dtype, 1.0, 1.0)

    def run_v1():
        for kc, vc in caches:
            ops.paged_attention_v1(o, q, kc, vc, Hkv, D ** -0.5, bt, sl, BS, ctx, None, kv_dtype, 1.0, 1.0)
    return run_rocm, run_v1, len(caches), nbytes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=32)
    ap.add_argument("--sweep", action="store_true")
    ap.add_argument("--fp8", action="store_true")
    ap.add_argument("--attn", action="store_true")
    args = ap.parse_args()
    res = []

    def report(tag, t, nbytes, extra=""):
        line = dict(kernel=tag, us=round(t * 1e6, 2), GBps=round(nbytes / t / 1e9, 1),
                    frac_of_8TBs=round(nbytes / t / 8e12, 3), cfg=extra)
        res.append(line)
        print(json.dumps(line), flush=True)

    for name in SHAPES:
        K, N = SHAPES[name]
        copies = max(2, (600 << 20) // (K * N // 2))
        run, n, nb = gptq_case(name, args.m, copies)
        for k in ("APHRO_WNA16_VEC", "APHRO_WNA16_KSPLIT", "APHRO_WNA16_GENERIC"):
            os.environ.pop(k, None)
        report(f"gptq_{name}_M{args.m}", timeit(run, n), nb, "default")
        if args.sweep:
            for nwv in (4, 8):
                for ps in (0, 1):
                    os.environ["APHRO_WNA16_NWV"] = str(nwv)
                    os.environ["APHRO_WNA16_PRESCALE"] = str(ps)
                    report(f"gptq_{name}_M{args.m}", timeit(run, n), nb, f"nwv{nwv} prescale{ps}")
            os.environ.pop("APHRO_WNA16_NWV", None)
            os.environ.pop("APHRO_WNA16_PRESCALE", None)
            for vec in (2, 4):
                for ks in (1, 2, 4):
                    os.environ["APHRO_WNA16_VEC"] = str(vec)
                    os.environ["APHRO_WNA16_KSPLIT"] = str(ks)
                    report(f"gptq_{name}_M{args.m}", timeit(run, n), nb, f"vec{vec} ksplit{ks}")
            os.environ.pop("APHRO_WNA16_VEC", None)
            os.environ.pop("APHRO_WNA16_KSPLIT", None)
            for dbg in (1, 2, 3, 4):
                os.environ["APHRO_WNA16_DBG"] = str(dbg)
                report(f"gptq_{name}_M{args.m}", timeit(run, n), nb, f"DEBUG dbg={dbg} (1: lane-linear A, 2: no A refill)")
            os.environ.pop("APHRO_WNA16_DBG", None)
            os.environ["APHRO_WNA16_GENERIC"] = "1"
            report(f"gptq_{name}_M{args.m}", timeit(run, n), nb, "generic kernel")
            os.environ.pop("APHRO_WNA16_GENERIC", None)
        del run
        torch.cuda.empty_cache()
    if args.fp8:
        for name in SHAPES:
            K, N = SHAPES[name]
            copies = max(2, (600 << 20) // (K * N))
            run, n, nb = fp8_case(name, args.m, copies)
            report(f"fp8_{name}_M{args.m}", timeit(run, n), nb, "default")
            if args.sweep:
                for nt in (2, 4):
                    for ks in (1, 2, 4):
                        os.environ["APHRO_FP8_NT"] = str(nt)
                        os.environ["APHRO_FP8_KSPLIT"] = str(ks)
                        report(f"fp8_{name}_M{args.m}", timeit(run, n), nb, f"nt{nt} ksplit{ks}")
                os.environ.pop("APHRO_FP8_NT", None)
                os.environ.pop("APHRO_FP8_KSPLIT", None)
            del run
            torch.cuda.empty_cache()
    if args.attn:
        for ctx in (512, 1024, 4096):
            for kvd in ("auto", "fp8"):
                layers = max(2, (600 << 20) // (2 * 32 * ctx * 8 * 128 * (2 if kvd == "auto" else 1)))
                r_rocm, r_v1, n, nb = attn_case(32, ctx, kvd, min(layers, 16))
                report(f"paged_attn_rocm_bs32_ctx{ctx}_{kvd}", timeit(r_rocm, n), nb)
                report(f"paged_attn_v1_bs32_ctx{ctx}_{kvd}", timeit(r_v1, n), nb)
                torch.cuda.empty_cache()
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/kbench.json", "w"), indent=1)


if __name__ == "__main__":
    main()
