#!/usr/bin/env python3
"""Launches each decode-step kernel a few times at the bench shapes (for rocprofv3 --pmc).  Round 3: gate_up runs the
resident kernel on strip-major weights, as the decode step does; the 33..64-row forms run at M = 64.  Round 4: the resident
entry point dispatches to the single-pass stream kernel for the four configs[1] plans (o_proj included); the FP8 W8A8
resident kernel runs on the same shapes at the end."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aphrodite_engine_amd import _custom_ops as ops  # noqa: E402
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
bs = 32


def gptq(k, n):
    qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (k // 8, n), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
    qz = torch.randint(-2 ** 31, 2 ** 31 - 1, (k // 128, n // 8), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
    sc = (torch.rand(k // 128, n, generator=g, device=dev) * 0.01).half()
    return qw, qz, sc


shapes = {"qkv": (4096, 6144), "o": (4096, 4096), "gate_up": (4096, 28672), "down": (14336, 4096)}
ws = {n: [gptq(*s) for _ in range(6)] for n, s in shapes.items()}
for name, (k, n) in shapes.items():
    a = ops.wna16_pack_a(torch.randn(bs, k, device=dev, dtype=torch.float16))
    a64 = ops.wna16_pack_a(torch.randn(64, k, device=dev, dtype=torch.float16))
    for qw, qz, sc in ws[name]:
        if name == "gate_up":
            ops.wna16_gemm_silu_pack(a, bs, k, qw, qz, sc, 1)                     # round-2 kernel (M <= 32 fallback)
            strip = ops.wna16_strip_relayout(qw, bs, k // 128)
            ops.wna16_gemm_resident(a, bs, k, strip, qz, sc, 1, mode="silu", strip_layout=True)   # what the step launches
            ops.wna16_gemm_mid_silu_pack(a64, 64, k, qw, qz, sc, 1)               # bs 64
        else:
            ops.wna16_gemm_packed(a, bs, k, qw, qz, sc, 1, partials=True)       # round-2 kernel (o_proj in the step; fallback elsewhere)
            if name in ("down", "qkv", "o"):                                     # what the step launches at <= 32 rows (round 4: o too)
                strip = ops.wna16_strip_relayout(qw, bs, k // 128)
                ops.wna16_gemm_resident(a, bs, k, strip, qz, sc, 1, mode="slabs", strip_layout=True)
            if name == "down":
                ops.wna16_gemm_mid_packed(a64, 64, k, qw, qz, sc, 1, partials=True)
# fused attention + norms
ctx, Hq, Hkv, D, BS = 1100, 32, 8, 128, 16
bps = (ctx + BS - 1) // BS
nb = bs * bps
bt = torch.randperm(nb, device=dev).view(bs, bps).int()
sl = torch.full((bs, ), ctx, dtype=torch.int32, device=dev)
slabs = torch.randn(2, bs, (Hq + 2 * Hkv) * D, device=dev) * 0.3
cs = torch.randn(bs, D, device=dev).half()
slots = (bt[:, (ctx - 1) // BS].long() * BS + (ctx - 1) % BS)
for _ in range(6):
    kc = torch.randn(nb, Hkv, D // 8, BS, 8, device=dev, dtype=torch.float16) * 0.1
    vc = torch.randn(nb, Hkv, D, BS, device=dev, dtype=torch.float16) * 0.1
    ops.paged_attention_rope_packed(slabs, None, cs, slots, kc, vc, Hq, Hkv, D ** -0.5, bt, sl, BS, ctx, None,
                                    "auto", 1.0, 1.0)
res = torch.randn(bs, 4096, device=dev).half()
w = torch.ones(4096, device=dev).half()
s4 = torch.randn(4, bs, 4096, device=dev)
for _ in range(6):
    ops.fused_add_rms_norm_pack(None, s4, res, True, w, 1e-5)
torch.cuda.synchronize()
# round 3, late: the LM head with the argmax folded in, and the op-level one-launch GEMMs (row-major activations)
V = 128256
lmw = [(torch.randn(V, 4096, device=dev, generator=g) * 0.02).half() for _ in range(2)]
hid = (torch.randn(bs, 4096, device=dev, generator=g) * 0.5).half()
for i in range(6):
    ops.lm_head_argmax(hid, lmw[i % 2], V)
gi = torch.empty(0, dtype=torch.int32, device=dev)
for name, (k, n) in shapes.items():
    x = torch.randn(bs, k, device=dev, dtype=torch.float16)
    for qw, qz, sc in ws[name]:
        ops.gptq_gemm(x, qw, qz, sc, gi, True, 4)
torch.cuda.synchronize()

# round 4: FP8 W8A8 resident decode GEMM (csrc/fp8_gemm_resident.hip) on the four projection shapes, slab form
for name, (k, n) in shapes.items():
    qa = torch.randint(0, 0x48, (bs, k), generator=g, device=dev, dtype=torch.int16).to(torch.uint8).view(torch.float8_e4m3fn)
    for _ in range(4):
        w8 = torch.randint(0, 0x48, (n, k), generator=g, device=dev, dtype=torch.int16).to(torch.uint8).view(torch.float8_e4m3fn)
        st8 = ops.fp8_strip_relayout(w8, bs)
        ops.fp8_gemm_resident(qa, st8, slabs=True)
torch.cuda.synchronize()

# round 6: the step's forms of the dynamic per-token scheme -- gate_up with SiluAndMul + absmax partials in the epilogue on the
# interleaved strip copy (same kernel template as the plain form: only this form is launched for it now), o_proj / down fed with
# pair-major 16-bit activations + partials and quantising on load (template <..., 1>)
for name, (k, n) in shapes.items():
    x16 = torch.randn(bs, k, device=dev, dtype=torch.float16, generator=g)
    if name in ("o", "down"):
        np_ = 8 if name == "o" else 256
        part = x16.float().abs().view(bs, np_, -1).amax(2).contiguous()
        xp = torch.zeros(ops.aq_pairs_numel(bs, k), dtype=x16.dtype, device=dev)
        xp[ops.aq_pairs_index(bs, k, dev).flatten()] = x16.flatten()
    for _ in range(4):
        w8 = torch.randint(0, 0x48, (n, k), generator=g, device=dev, dtype=torch.int16).to(torch.uint8).view(torch.float8_e4m3fn)
        if name == "gate_up":
            qa, sa = ops.scaled_fp8_quant(x16, None, use_per_token_if_dynamic=True)
            ops.fp8_gemm_resident_silu(qa, ops.fp8_strip_relayout_interleaved(w8, bs), sa, torch.full((n, 1), 0.01, device=dev), torch.float16,
                                       act_pairs=True)
        elif name in ("o", "down"):
            ops.fp8_gemm_resident_aq(xp, part, ops.fp8_strip_relayout(w8, bs), a_pairs=True)
torch.cuda.synchronize()
