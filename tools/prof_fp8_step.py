#!/usr/bin/env python3
"""A few eager decode steps of a 6-layer Llama-3-8B-shaped model in one quant format at the bench
shape (bs 32, ctx 1024), for `rocprofv3 --kernel-trace --stats`: per-kernel durations of the step.
usage: prof_fp8_step.py [fp8ct|fp8|gptq] [layers] [steps]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aphrodite_engine_amd import model as M  # noqa: E402
from aphrodite_engine_amd.quantization.fp8 import CompressedTensorsW8A8Fp8Config, Fp8Config  # noqa: E402
from aphrodite_engine_amd.quantization.gptq import GPTQConfig  # noqa: E402

quant = sys.argv[1] if len(sys.argv) > 1 else "fp8ct"
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 6
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
qc = {"fp8ct": CompressedTensorsW8A8Fp8Config("channel"), "fp8": Fp8Config(True, "dynamic"),
      "gptq": GPTQConfig(4, 128, False)}[quant]
dev = torch.device("cuda")
cfg = M.LlamaConfig(num_hidden_layers=layers)
with torch.no_grad():
    m = M.LlamaForCausalLM(cfg, qc, torch.float16).init_synthetic(dev)
    if quant == "gptq":
        for l in m.layers:
            l.enable_fused_silu(32)
    bs, ctx = 32, 1024
    meta, pos, nblocks = M.make_decode_metadata(bs, ctx, 16, dev)
    caches = M.make_kv_caches(cfg, nblocks, 16, torch.float16, "auto", dev)
    ids = torch.randint(0, cfg.vocab_size, (bs, ), device=dev)
    for _ in range(steps):
        h = m(ids, pos, caches, meta)
    torch.cuda.synchronize()
print("done", quant, layers, steps)
