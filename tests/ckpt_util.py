"""Synthetic Hugging Face checkpoints in the on-disk formats the loader ingests (test input
generator).  Weights are quantised / packed with the oracle's restated reference packers
(oracle/quant.py: quant_utils.py:123-441); every tensor written is also returned, together with
the LOGICAL integer / float matrices, so tests can derive each rank's expected shard by plain
index arithmetic that shares no code with aphrodite_engine_amd/loader.py."""
import json
import os

import numpy as np
import torch
from safetensors.torch import save_file

from oracle import quant as oq

PROJS = {  # hf projection name -> (in_features attr, out_features attr) resolved in _dims
    "self_attn.q_proj": ("h", "q"), "self_attn.k_proj": ("h", "kv"), "self_attn.v_proj": ("h", "kv"),
    "self_attn.o_proj": ("q", "h"), "mlp.gate_proj": ("h", "i"), "mlp.up_proj": ("h", "i"),
    "mlp.down_proj": ("i", "h")}


def _dims(cfg):
    hd = cfg.hidden_size // cfg.num_attention_heads
    return {"h": cfg.hidden_size, "q": cfg.num_attention_heads * hd, "kv": cfg.num_key_value_heads * hd,
            "i": cfg.intermediate_size}


def hf_config(cfg, extra=None):
    d = {"architectures": ["LlamaForCausalLM"], "model_type": "llama", "hidden_size": cfg.hidden_size,
         "intermediate_size": cfg.intermediate_size, "num_hidden_layers": cfg.num_hidden_layers,
         "num_attention_heads": cfg.num_attention_heads, "num_key_value_heads": cfg.num_key_value_heads,
         "vocab_size": cfg.vocab_size, "rms_norm_eps": cfg.rms_norm_eps, "rope_theta": cfg.rope_theta,
         "max_position_embeddings": cfg.max_position_embeddings, "tie_word_embeddings": False}
    if cfg.num_local_experts:
        d.update({"architectures": ["MixtralForCausalLM"], "model_type": "mixtral",
                  "num_local_experts": cfg.num_local_experts, "num_experts_per_tok": cfg.num_experts_per_tok})
    d.update(extra or {})
    return d


def fp8_quant(w, per_channel):
    """w float32 [N, K] -> (e4m3 tensor [N, K], scale)."""
    wt = torch.from_numpy(w)
    if per_channel:
        s = (wt.abs().amax(dim=1, keepdim=True) / 448.0).float()
    else:
        s = (wt.abs().max() / 448.0).float()
    return (wt / s).clamp(-448, 448).to(torch.float8_e4m3fn), s


def write_checkpoint(path, cfg, fmt, seed=0, group_size=128, kv_scales=None, fp8_static=False,
                     embedded_config=True, fused_on_disk=False, extra_bias=True, bits=4, proj_bias=None):
    """fmt: fp16 | gptq | awq | fp8 | ct-fp8-channel | ct-fp8-tensor | ct-w8a16 | ct-w4a16 | ct-w8a16i (int8 pack-quantized).
    kv_scales: None | "kv" (per-layer k_scale + v_scale) | "legacy" (kv_scale) | "ct" ({k,v}_proj.output_scale).
    proj_bias: None | "attn" | "mlp" | "both" -- real projection biases (config.attention_bias / mlp_bias), also in
    logical[name]["bias"].
    Returns {"tensors": {hf name: tensor}, "logical": {hf module name: dict of logical matrices}}."""
    os.makedirs(path, exist_ok=True)
    rng = np.random.default_rng(seed)
    dims = _dims(cfg)
    tensors, logical = {}, {}
    qcfg, extra_files = None, {}
    tensors["model.embed_tokens.weight"] = torch.from_numpy(
        rng.standard_normal((cfg.vocab_size, cfg.hidden_size)).astype(np.float32)).half()
    tensors["model.norm.weight"] = torch.from_numpy(rng.random(cfg.hidden_size).astype(np.float32) + 0.5).half()
    tensors["lm_head.weight"] = torch.from_numpy(
        (rng.standard_normal((cfg.vocab_size, cfg.hidden_size)) * 0.05).astype(np.float32)).half()
    for li in range(cfg.num_hidden_layers):
        base = f"model.layers.{li}."
        for nm in ("input_layernorm", "post_attention_layernorm"):
            tensors[base + nm + ".weight"] = torch.from_numpy(
                rng.random(cfg.hidden_size).astype(np.float32) + 0.5).half()
        tensors[base + "self_attn.rotary_emb.inv_freq"] = torch.zeros(4)      # must be skipped
        projs = dict(PROJS)
        if cfg.num_local_experts:     # Mixtral: sparse MLP -- router + experts.{e}.w1 (gate) / w3 (up) / w2 (down)
            for p in ("mlp.gate_proj", "mlp.up_proj", "mlp.down_proj"):
                del projs[p]
            for e in range(cfg.num_local_experts):
                projs[f"block_sparse_moe.experts.{e}.w1"] = ("h", "i")
                projs[f"block_sparse_moe.experts.{e}.w3"] = ("h", "i")
                projs[f"block_sparse_moe.experts.{e}.w2"] = ("i", "h")
            tensors[base + "block_sparse_moe.gate.weight"] = torch.from_numpy(
                (rng.standard_normal((cfg.num_local_experts, cfg.hidden_size)) * 0.1).astype(np.float32)).half()
        for proj, (ki, ni) in projs.items():
            K, N = dims[ki], dims[ni]
            w = (rng.standard_normal((K, N)) * 0.05).astype(np.float32)        # [K, N] = weight.T
            name = base + proj
            if fmt == "fp16":
                tensors[name + ".weight"] = torch.from_numpy(np.ascontiguousarray(w.T)).half()
                logical[name] = {"w": w.astype(np.float16)}
            elif fmt in ("gptq", "awq"):
                _, q, s, zp = oq.quantize_weights(w, bits if fmt == "gptq" else 4, group_size, zero_points=True)
                logical[name] = {"q": q, "s": s.astype(np.float16), "zp": zp}
                if fmt == "gptq":
                    tensors[name + ".qweight"] = torch.from_numpy(oq.gptq_pack(q, bits).astype(np.int32))
                    tensors[name + ".qzeros"] = torch.from_numpy(oq.gptq_pack_zeros(zp, bits).astype(np.int32))
                    tensors[name + ".g_idx"] = torch.from_numpy((np.arange(K) // group_size).astype(np.int32))
                    if extra_bias:
                        tensors[name + ".bias"] = torch.zeros(N, dtype=torch.float16)   # AutoGPTQ exports these
                else:
                    tensors[name + ".qweight"] = torch.from_numpy(oq.awq_pack(q, 4).astype(np.int32))
                    tensors[name + ".qzeros"] = torch.from_numpy(oq.awq_pack(zp, 4).astype(np.int32))
                tensors[name + ".scales"] = torch.from_numpy(s.astype(np.float16))
            elif fmt in ("fp8", "ct-fp8-channel", "ct-fp8-tensor", "ct-w8a16"):
                per_channel = fmt in ("ct-fp8-channel", "ct-w8a16")
                wq, s = fp8_quant(np.ascontiguousarray(w.T), per_channel)
                tensors[name + ".weight"] = wq
                if fmt == "fp8":
                    tensors[name + ".weight_scale"] = s.reshape(())            # AutoFP8: 0-dim
                elif per_channel:
                    tensors[name + ".weight_scale"] = s.reshape(N, 1)
                else:
                    tensors[name + ".weight_scale"] = s.reshape(1)             # compressed-tensors: [1]
                if fp8_static:
                    iscale = torch.tensor(0.01 + 0.001 * rng.random(), dtype=torch.float32)
                    tensors[name + ".input_scale"] = iscale.reshape(()) if fmt == "fp8" else iscale.reshape(1)
                logical[name] = {"wq": wq, "s": s, "input_scale": tensors.get(name + ".input_scale")}
            elif fmt in ("ct-w4a16", "ct-w8a16i"):
                wbits = 4 if fmt == "ct-w4a16" else 8
                _, q, s, _ = oq.quantize_weights(w, wbits, group_size, zero_points=False)
                logical[name] = {"q": q, "s": s.astype(np.float16)}
                tensors[name + ".weight_packed"] = torch.from_numpy(
                    np.ascontiguousarray(oq.gptq_pack(q, wbits).T).astype(np.int32))           # [N, K/8] ([N, K/4])
                tensors[name + ".weight_scale"] = torch.from_numpy(np.ascontiguousarray(s.T).astype(np.float16))
                tensors[name + ".weight_shape"] = torch.tensor([N, K], dtype=torch.int64)
            else:
                raise ValueError(fmt)
            if proj_bias in ("both", "attn" if proj.startswith("self_attn") else "mlp") and "experts" not in proj:
                b = (rng.standard_normal(N) * 0.5).astype(np.float16)
                tensors[name + ".bias"] = torch.from_numpy(b)
                logical[name]["bias"] = b
        if kv_scales == "kv":
            tensors[base + "self_attn.k_scale"] = torch.tensor(0.02 + 0.001 * li, dtype=torch.float32)
            tensors[base + "self_attn.v_scale"] = torch.tensor(0.03 + 0.001 * li, dtype=torch.float32)
        elif kv_scales == "legacy":
            tensors[base + "self_attn.kv_scale"] = torch.tensor(0.05 + 0.001 * li, dtype=torch.float32)
        elif kv_scales == "ct":
            tensors[base + "self_attn.k_proj.output_scale"] = torch.tensor([0.02 + 0.001 * li])
            tensors[base + "self_attn.v_proj.output_scale"] = torch.tensor([0.03 + 0.001 * li])
    if fused_on_disk:   # Phi-3 style: qkv_proj / gate_up_proj stored fused (fp16 only)
        assert fmt == "fp16"
        for li in range(cfg.num_hidden_layers):
            base = f"model.layers.{li}."
            qkv = [tensors.pop(base + f"self_attn.{p}_proj.weight") for p in "qkv"]
            tensors[base + "self_attn.qkv_proj.weight"] = torch.cat(qkv, 0)
            gu = [tensors.pop(base + f"mlp.{p}_proj.weight") for p in ("gate", "up")]
            tensors[base + "mlp.gate_up_proj.weight"] = torch.cat(gu, 0)
    if fmt == "gptq":
        qcfg = {"bits": bits, "group_size": group_size, "desc_act": False, "quant_method": "gptq"}
        if not embedded_config:
            extra_files["quantize_config.json"] = {k: v for k, v in qcfg.items() if k != "quant_method"}
    elif fmt == "awq":
        qcfg = {"quant_method": "awq", "bits": 4, "group_size": group_size, "zero_point": True, "version": "gemm"}
    elif fmt == "fp8":
        qcfg = {"quant_method": "fp8", "activation_scheme": "static" if fp8_static else "dynamic",
                "ignored_layers": ["lm_head"]}
    elif fmt.startswith("ct-"):
        weights = {"ct-fp8-channel": {"num_bits": 8, "type": "float", "symmetric": True, "strategy": "channel",
                                      "dynamic": False},
                   "ct-fp8-tensor": {"num_bits": 8, "type": "float", "symmetric": True, "strategy": "tensor",
                                     "dynamic": False},
                   "ct-w8a16": {"num_bits": 8, "type": "float", "symmetric": True, "strategy": "channel",
                                "dynamic": False},
                   "ct-w4a16": {"num_bits": 4, "type": "int", "symmetric": True, "strategy": "group",
                                "group_size": group_size, "dynamic": False},
                   "ct-w8a16i": {"num_bits": 8, "type": "int", "symmetric": True, "strategy": "group",
                                 "group_size": group_size, "dynamic": False}}[fmt]
        acts = None
        if fmt in ("ct-fp8-channel", "ct-fp8-tensor"):
            acts = ({"num_bits": 8, "type": "float", "symmetric": True, "strategy": "tensor", "dynamic": False}
                    if fp8_static else
                    {"num_bits": 8, "type": "float", "symmetric": True, "strategy": "token", "dynamic": True})
        qcfg = {"quant_method": "compressed-tensors",
                "format": "pack-quantized" if fmt in ("ct-w4a16", "ct-w8a16i") else "float-quantized",
                "config_groups": {"group_0": {"targets": ["Linear"], "weights": weights,
                                              "input_activations": acts}},
                "ignore": ["lm_head"]}
    hf = hf_config(cfg, {"quantization_config": qcfg} if (qcfg and embedded_config) else None)
    if proj_bias in ("attn", "both"):
        hf["attention_bias"] = True
    if proj_bias in ("mlp", "both"):
        hf["mlp_bias"] = True
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(hf, f)
    for fname, doc in extra_files.items():
        with open(os.path.join(path, fname), "w") as f:
            json.dump(doc, f)
    # two shard files + an index, like a real multi-file export
    names = sorted(tensors)
    half = len(names) // 2
    parts = {"model-00001-of-00002.safetensors": names[:half], "model-00002-of-00002.safetensors": names[half:]}
    weight_map = {}
    for fname, ns in parts.items():
        save_file({n: tensors[n].contiguous() for n in ns}, os.path.join(path, fname))
        weight_map.update({n: fname for n in ns})
    with open(os.path.join(path, "model.safetensors.index.json"), "w") as f:
        json.dump({"metadata": {}, "weight_map": weight_map}, f)
    return {"tensors": tensors, "logical": logical}
