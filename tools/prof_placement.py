#!/usr/bin/env python3
"""Does the placement effect of profiles/r6_one_copy.txt (4) show in the address-translation counters?  Under
`rocprofv3 --pmc ... --kernel-trace`: a 32-layer Llama-3-8B int4 model with one resident copy; the gate_up / down / qkv / o decode
GEMMs over all layers (phase A: the copies where aphro_wna16_strip_relayout first wrote them), then every layer's layouts
released and rebuilt (restore_op_level_layouts + enable_fused_silu + enable_one_copy: the churn of bench.py's op-by-op leg),
then the same launches again (phase B).  tools/prof_placement_reduce.py splits the dispatches at the last relayout launch."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aphrodite_engine_amd import _custom_ops as ops  # noqa: E402
from aphrodite_engine_amd import model as Mo  # noqa: E402
from aphrodite_engine_amd.quantization.gptq import GPTQConfig  # noqa: E402

DEV = "cuda:0"
bs = 32


def build_layouts(m):
    for layer in m.layers:
        layer.enable_fused_silu(bs, keep_original=False)
    for layer in m.layers:
        layer.enable_one_copy()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()


def launches(m, reps=3):
    h, inter = m.cfg.hidden_size, m.cfg.intermediate_size
    px = ops.wna16_pack_a(torch.randn(bs, h, device=DEV, dtype=torch.float16))
    pd = ops.wna16_pack_a(torch.randn(bs, inter, device=DEV, dtype=torch.float16))
    for _ in range(reps):
        for layer in m.layers:
            _, qz, sc, zo = layer.gate_up_interleaved
            ops.wna16_gemm_resident(px, bs, h, layer.gate_up_strip, qz, sc, zo, mode="silu", strip_layout=True)
        for layer in m.layers:
            layer._gemm_slabs("down_proj", pd, bs, inter)
        for layer in m.layers:
            layer._gemm_slabs("qkv_proj", px, bs, h)
        for layer in m.layers:
            layer._gemm_slabs("o_proj", px, bs, h)
    torch.cuda.synchronize()


def main():
    with torch.no_grad():
        m = Mo.LlamaForCausalLM(Mo.LLAMA3_8B, GPTQConfig(4, 128, False), torch.float16, "auto").init_synthetic(DEV, seed=0)
        build_layouts(m)
        launches(m)                     # phase A
        for layer in m.layers:
            layer.restore_op_level_layouts()
        build_layouts(m)                # the churn (relayout launches: the phase boundary)
        launches(m)                     # phase B


if __name__ == "__main__":
    main()
