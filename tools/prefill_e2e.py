"""End-to-end prefill (time to first token) of Llama-3-8B through this package's model: one 8192-token prompt (BASELINE
configs[2]'s sequence length), GPTQ int4 g128 and compressed-tensors FP8 W8A8, fp16 / fp8 KV cache.  Everything the
prefill step launches is on the clock: embedding gather, 32 x (norm, qkv GEMM, rotary, cache write, causal attention,
o GEMM, norm, gate_up GEMM, SiluAndMul, down GEMM), final norm, lm_head of the last token, argmax.
    python tools/prefill_e2e.py [T]"""
import dataclasses
import json
import sys
import time

import torch

from aphrodite_engine_amd import model as M
from aphrodite_engine_amd.attention.backend import MI355XAttentionMetadata
from aphrodite_engine_amd.quantization.gptq import GPTQConfig

T = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device("cuda:0")
BS = 16


def run(name, qc, kv):
    cfg = dataclasses.replace(M.LLAMA3_8B, max_position_embeddings=max(8192, T))
    with torch.no_grad():
        model = M.LlamaForCausalLM(cfg, qc, torch.float16, kv)
        model.init_synthetic(dev, seed=0)
        nblk = (T + BS - 1) // BS
        caches = M.make_kv_caches(cfg, nblk, BS, torch.float16, kv, dev, fill=False)
        bt = torch.randperm(nblk, device=dev).to(torch.int32).view(1, nblk)
        pos = torch.arange(T, device=dev, dtype=torch.int64)
        slots = (bt[0, (pos // BS)].long() * BS + pos % BS)
        meta = MI355XAttentionMetadata(
            num_prefills=1, num_prefill_tokens=T, num_decode_tokens=0, slot_mapping=slots, seq_lens=[T],
            seq_lens_tensor=torch.tensor([T], dtype=torch.int32, device=dev), max_query_len=T, max_prefill_seq_len=T,
            max_decode_seq_len=0, query_start_loc=torch.tensor([0, T], dtype=torch.int32, device=dev),
            seq_start_loc=torch.tensor([0, T], dtype=torch.int32, device=dev),
            context_lens_tensor=torch.zeros(1, dtype=torch.int32, device=dev), block_tables=bt, use_cuda_graph=False)
        ids = torch.randint(0, cfg.vocab_size, (T, ), device=dev)

        def step():
            h = model(ids, pos, caches, meta)
            return model.sample_greedy(model.compute_logits(h[-1:]))
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 3
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
    # dense-layer flops of the prompt (GEMMs + causal attention)
    h, i = cfg.hidden_size, cfg.intermediate_size
    qkv = h * (cfg.num_attention_heads + 2 * cfg.num_key_value_heads) * cfg.head_dim
    gemm = 2.0 * T * (qkv + h * h + 3 * h * i) * cfg.num_hidden_layers
    attn = 4.0 * T * T * cfg.head_dim * cfg.num_attention_heads / 2 * cfg.num_hidden_layers
    print(json.dumps({"prefill": name, "kv_cache": kv, "tokens": T, "ms": round(dt * 1e3, 2), "tokens_per_s": round(T / dt),
                      "TFLOPs": round((gemm + attn) / dt / 1e12, 1)}), flush=True)
    del model, caches
    torch.cuda.empty_cache()


run("Llama-3-8B GPTQ int4 g128", GPTQConfig(4, 128, False), "auto")
from aphrodite_engine_amd.quantization.fp8 import CompressedTensorsW8A8Fp8Config  # noqa: E402
run("Llama-3-8B compressed-tensors FP8 W8A8 (per-token x per-channel)", CompressedTensorsW8A8Fp8Config(strategy="channel", is_static_input_scheme=False), "fp8")
