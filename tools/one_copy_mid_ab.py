#!/usr/bin/env python3
"""33..64-row decode on a model whose layouts were built for <= 32 rows (what the engine adapter builds), with two copies of
the int4 matrices and with one (round 6): 8 layers of Llama-3-8B geometry, graph-captured step, ms per step.
  two copies : MLP weights on the one-pass kernel over [K/8, N], qkv / o on the round-2 kernel over [K/8, N]
  one copy   : MLP weights on the one-pass kernel over the strip-major copy (strip_m), qkv / o on two 32-row halves of the stream kernel
  one copy, APHRO_DECODE_ROW_HALVES=1 : everything on the halves
usage: python tools/one_copy_mid_ab.py   -> profiles/r6_one_copy.txt (5)"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aphrodite_engine_amd import model as Mo  # noqa: E402
from aphrodite_engine_amd.quantization.gptq import GPTQConfig  # noqa: E402

DEV = "cuda:0"


def build(one):
    cfg = Mo.LlamaConfig(hidden_size=4096, intermediate_size=14336, num_hidden_layers=8, num_attention_heads=32,
                         num_key_value_heads=8, vocab_size=1024, max_position_embeddings=2048)
    m = Mo.LlamaForCausalLM(cfg, GPTQConfig(4, 128, False), torch.float16, "auto").init_synthetic(DEV, seed=2)
    for layer in m.layers:
        layer.enable_fused_silu(32, keep_original=False)
    if one:
        for layer in m.layers:
            layer.enable_one_copy()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    return cfg, m


def step_ms(cfg, m, bs, ctx=1024):
    meta, pos, nblocks = Mo.make_decode_metadata(bs, ctx, 16, DEV)
    kv = Mo.make_kv_caches(cfg, nblocks, 16, torch.float16, "auto", DEV, seed=3)
    ids = torch.arange(bs, device=DEV) % cfg.vocab_size
    assert all(l.fused_decode_ok(bs) for l in m.layers)
    with torch.no_grad():
        out = m(ids, pos, kv, meta)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = m(ids, pos, kv, meta)
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(3):
            e0.record()
            for _ in range(20):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20)
    return best, out.float().argmax(-1)


def kernels():
    """The MLP GEMMs of one layer at 48 / 64 rows: one-pass kernel on [K/8, N], on the strip-major copy, and the stream kernel's
    two 32-row halves on the strip-major copy (us per launch, 16 distinct weight sets cycled, graph-captured)."""
    from aphrodite_engine_amd import _custom_ops as ops
    g = torch.Generator(device=DEV).manual_seed(0)
    print("MLP GEMMs at 33..64 rows (us per launch; 16 weight sets cycled)")
    for name, K, N, silu in (("gate_up", 4096, 28672, True), ("down", 14336, 4096, False)):
        sets = []
        for _ in range(16):
            qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // 8, N), generator=g, device=DEV, dtype=torch.int64).to(torch.int32)
            qz = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // 128, N // 8), generator=g, device=DEV, dtype=torch.int64).to(torch.int32)
            sc = (torch.rand(K // 128, N, generator=g, device=DEV) * 0.01 + 0.005).half()
            sets.append((qw, ops.wna16_strip_relayout(qw, 32, K // 128), qz, sc))
        for M in (48, 64):
            packed = ops.wna16_pack_a((torch.randn(M, K, generator=g, device=DEV) * 0.5).half())

            def mid(strip):
                for qw, st, qz, sc in sets:
                    if silu:
                        ops.wna16_gemm_mid_silu_pack(packed, M, K, st if strip else qw, qz, sc, 1, strip_m=32 if strip else 0)
                    else:
                        ops.wna16_gemm_mid_packed(packed, M, K, st if strip else qw, qz, sc, 1, partials=True, strip_m=32 if strip else 0)

            def halves():
                for qw, st, qz, sc in sets:
                    ops.wna16_gemm_resident(packed, M, K, st, qz, sc, 1, mode="silu" if silu else "slabs", strip_layout=True)

            row = []
            for fn in (lambda: mid(False), lambda: mid(True), halves):
                fn()
                torch.cuda.synchronize()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr):
                    fn()
                gr.replay()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                best = 1e9
                for _ in range(3):
                    e0.record()
                    for _ in range(5):
                        gr.replay()
                    e1.record()
                    torch.cuda.synchronize()
                    best = min(best, e0.elapsed_time(e1) / (5 * len(sets)) * 1e3)
                row.append(best)
            print(f"  {name:>8} M={M}: one-pass [K/8, N] {row[0]:6.2f}   one-pass strip-major {row[1]:6.2f}   two halves {row[2]:6.2f}")
        del sets
        torch.cuda.empty_cache()


def main():
    kernels()
    res = {}
    for mode in ("two", "one", "one+halves"):
        if mode == "one+halves":
            os.environ["APHRO_DECODE_ROW_HALVES"] = "1"
        cfg, m = build(mode != "two")
        os.environ.pop("APHRO_DECODE_ROW_HALVES", None) if mode != "one+halves" else None
        for bs in (48, 64):
            res[(mode, bs)] = step_ms(cfg, m, bs)
        os.environ.pop("APHRO_DECODE_ROW_HALVES", None)
        del m
        torch.cuda.empty_cache()
    print("8 layers of Llama-3-8B geometry, layouts built for <= 32 rows, decode at 48 / 64 rows, ctx 1024 (ms per step)")
    for bs in (48, 64):
        t2 = res[("two", bs)][0]
        for mode in ("two", "one", "one+halves"):
            t, tok = res[(mode, bs)]
            same = bool(torch.equal(tok, res[("two", bs)][1]))
            print(f"  bs {bs:>2} {mode:>11}: {t:.4f} ms  ({t / t2:.3f} x two copies; greedy tokens equal: {same})")


if __name__ == "__main__":
    main()
