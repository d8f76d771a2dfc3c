#!/usr/bin/env python3
"""Round-3 lab: (A) the decode GEMMs / attention with their operands WARM in the 256 MiB Infinity Cache vs cold
(cycling over > 600 MB), (B) the whole configs[1] decode step (HIP graph) with weight-prefetch plans on a side stream.
Part B needs model.WeightPrefetcher, which exists at commit 4299031 only (measured: every plan made the step SLOWER, 3.10 ->
4.27 .. 4.99 ms, profiles/r3_prefetch_lab.txt; the hooks were removed from the product model again) -- `--parts A` runs here.
Prints one JSON line per measurement; everything lands in gpurun_out/prefetch_lab.jsonl."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from aphrodite_engine_amd import _custom_ops as ops  # noqa: E402

OUT = []


def emit(**kw):
    OUT.append(kw)
    print(json.dumps(kw), flush=True)


def timeit(fn, n_launch, iters=10):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        g.replay()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) * 1e-3 / (iters * n_launch)


SHAPES = {"qkv": (4096, 6144), "o": (4096, 4096), "gate_up": (4096, 28672), "down": (14336, 4096)}


def part_a(M=32):
    g = torch.Generator(device="cuda").manual_seed(0)
    for name, (K, N) in SHAPES.items():
        G = K // 128
        wbytes = K * N // 2
        ncold = max(2, (640 << 20) // wbytes)
        nwarm = max(1, (120 << 20) // wbytes)      # > the 32 MiB of L2, < the 256 MiB Infinity Cache
        ws = []
        for _ in range(ncold):
            qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // 8, N), generator=g, device="cuda", dtype=torch.int64).to(torch.int32)
            qz = torch.randint(-2 ** 31, 2 ** 31 - 1, (G, N // 8), generator=g, device="cuda", dtype=torch.int64).to(torch.int32)
            sc = (torch.rand(G, N, generator=g, device="cuda") * 0.01).half()
            ws.append((qw, qz, sc))
        a = torch.randn(M, K, device="cuda", dtype=torch.float16)
        packed = ops.wna16_pack_a(a)
        silu = name == "gate_up"

        def run(wl):
            for qw, qz, sc in wl:
                if silu:
                    ops.wna16_gemm_silu_pack(packed, M, K, qw, qz, sc, 1)
                else:
                    ops.wna16_gemm_packed(packed, M, K, qw, qz, sc, 1, partials=True)
        nb = wbytes + G * N * 2 + G * N // 2 + M * K * 2
        # the SAME number of launches per graph in every state (a graph replay has a fixed cost of a few us: comparing a
        # 1-launch graph with an 11-launch graph charges it to the wrong side -- the first version of this lab did)
        L = max(24, ncold)
        for tag, wl in (("cold", ws), ("mall_warm", ws[:nwarm]), ("one_copy", ws[:1])):
            seq = [wl[i % len(wl)] for i in range(L)]
            t = timeit(lambda: run(seq), L)
            emit(part="A", kernel=name, M=M, state=tag, copies=len(wl), launches_per_graph=L, us=round(t * 1e6, 2),
                 TBps=round(nb / t / 1e12, 3))
        del ws
        torch.cuda.empty_cache()
    # attention: one layer's KV (138 MB) warm vs 8 layers cold
    Hq, Hkv, D, BS, bs, ctx = 32, 8, 128, 16, 32, 1040
    bps = (ctx + BS - 1) // BS
    nb_ = bs * bps
    caches = []
    for _ in range(6):
        kc = torch.randn(nb_, Hkv, D // 8, BS, 8, device="cuda", dtype=torch.float16) * 0.1
        vc = torch.randn(nb_, Hkv, D, BS, device="cuda", dtype=torch.float16) * 0.1
        caches.append((kc, vc))
    bt = torch.randperm(nb_, device="cuda").view(bs, bps).int()
    sl = torch.full((bs, ), ctx, dtype=torch.int32, device="cuda")
    q = torch.randn(bs, Hq, D, device="cuda", dtype=torch.float16)

    def run_attn(cl):
        for kc, vc in cl:
            ops.paged_attention_packed(q, kc, vc, Hkv, D ** -0.5, bt, sl, BS, ctx, None, "auto", 1.0, 1.0)
    nbytes = 2 * bs * ctx * Hkv * D * 2
    for tag, cl in (("cold", caches), ("one_copy", caches[:1])):
        seq = [cl[i % len(cl)] for i in range(12)]
        t = timeit(lambda: run_attn(seq), 12)
        emit(part="A", kernel="paged_attention_packed", state=tag, copies=len(cl), launches_per_graph=12, us=round(t * 1e6, 2),
             TBps=round(nbytes / t / 1e12, 3))


def part_b(plans, steps=40, blocks=(512, )):
    ns = argparse.Namespace(gpus=1, steps=steps, warmup=8, model="llama3-8b", quant="gptq", kv_cache_dtype="auto",
                            batch=32, ctx=1024, parallelism="dp", sampling="greedy", no_overlap=False, no_graph=False,
                            no_cpu_baseline=True, no_prefill_info=True, layers=0)
    device = torch.device("cuda", 0)
    from aphrodite_engine_amd import distributed as D
    D.init_tensor_parallel(1)
    model, cfg, dtype = bench.build(ns, device)
    total = (steps + 12) * (len(plans) * len(blocks) + 2) + 8
    loop = bench.DecodeLoop(model, cfg, dtype, ns, device, total)
    ref_tokens = None
    with torch.no_grad():
        for _ in range(2):
            loop.step()
        torch.cuda.synchronize()
        for nb in blocks:
            os.environ["APHRO_PREFETCH_BLOCKS"] = str(nb)
            for plan in plans:
                pf = model.enable_weight_prefetch(device, plan)
                graph = torch.cuda.CUDAGraph()
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    loop.step()
                torch.cuda.current_stream().wait_stream(s)
                torch.cuda.synchronize()
                if pf is not None:
                    pf.bytes = 0
                with torch.cuda.graph(graph):
                    loop.step()
                for _ in range(8):
                    graph.replay()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    graph.replay()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / steps
                emit(part="B", plan=plan or "off", blocks=nb, ms_per_step=round(dt * 1e3, 4),
                     tokens_per_s=round(32 / dt, 1), prefetched_MB_per_step=round((pf.bytes if pf else 0) / 1e6, 1),
                     ctx=int(loop.meta.seq_lens_tensor[0].item()))
                del graph
    model.enable_weight_prefetch(device, None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--parts", default="A")
    ap.add_argument("--plans", default="")
    args = ap.parse_args()
    if "A" in args.parts:
        part_a()
        torch.cuda.empty_cache()
    if "B" in args.parts:
        plans = args.plans.split("|") if args.plans else [
            "", "P0:o,gate_up,down,qkv+1", "P0:o,gate_up;P2:down,qkv+1", "P2:gate_up,down,qkv+1",
            "P0:o;P2:gate_up,down,qkv+1", "P1:o,gate_up;P2:down,qkv+1", "P2:down,qkv+1", "P3:down,qkv+1",
            "P0:o,gate_up", "P0:gate_up", "P2:gate_up", ""]
        part_b(plans, blocks=(512, 256))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "prefetch_lab.jsonl"), "w") as f:
        for r in OUT:
            f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
