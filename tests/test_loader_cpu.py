"""On-disk format ingestion on CPU: synthetic Hugging Face checkpoints in every
supported format are loaded for tensor-parallel sizes 1, 2 and 4 (TINY has 2 KV heads, so 4 ranks
exercise KV-head replication) and every parameter is compared with the shard derived here by plain
index arithmetic on the LOGICAL matrices -- no code shared with the loader.  No post-processing
(repack / requantise) is run: those are GPU ops and are covered by the -m gpu tests."""
import json
import os

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from aphrodite_engine_amd import loader as L
from aphrodite_engine_amd import model as M
from aphrodite_engine_amd.distributed import simulated_tensor_parallel
from oracle import quant as oq
from tests import ckpt_util as CU

CFG = M.TINY   # hidden 512, 4 heads x 128, 2 KV heads, intermediate 1024, 2 layers
HD = CFG.hidden_size // CFG.num_attention_heads


# ----------------------------------------------------------------------------------------------
# expected shards, by index sets over the logical [K, N] matrices
# ----------------------------------------------------------------------------------------------
def col_sets(rank, world):
    """Logical output columns this rank holds, per fused layer, in parameter order."""
    hq, hkv, inter = CFG.num_attention_heads, CFG.num_key_value_heads, CFG.intermediate_size
    q_heads = list(range(rank * hq // world, (rank + 1) * hq // world))
    if world >= hkv:
        kv_heads = [rank // (world // hkv)]
    else:
        kv_heads = list(range(rank * hkv // world, (rank + 1) * hkv // world))

    def cols(heads):
        return np.concatenate([np.arange(h * HD, (h + 1) * HD) for h in heads])
    i0, i1 = rank * inter // world, (rank + 1) * inter // world
    return {"qkv": [("self_attn.q_proj", cols(q_heads)), ("self_attn.k_proj", cols(kv_heads)),
                    ("self_attn.v_proj", cols(kv_heads))],
            "gate_up": [("mlp.gate_proj", np.arange(i0, i1)), ("mlp.up_proj", np.arange(i0, i1))]}


def row_set(k, rank, world):
    return np.arange(rank * k // world, (rank + 1) * k // world)


def expected_logical(truth, layer, rank, world, key):
    """{module: logical matrix [K_local, N_local]} for qkv_proj, o_proj, gate_up_proj, down_proj."""
    base = f"model.layers.{layer}."
    lg = truth["logical"]
    sets = col_sets(rank, world)
    out = {}
    out["qkv_proj"] = np.concatenate([lg[base + p][key][:, c] for p, c in sets["qkv"]], axis=1)
    out["gate_up_proj"] = np.concatenate([lg[base + p][key][:, c] for p, c in sets["gate_up"]], axis=1)
    for mod, proj in (("o_proj", "self_attn.o_proj"), ("down_proj", "mlp.down_proj")):
        full = lg[base + proj][key]
        out[mod] = full[row_set(full.shape[0], rank, world), :]
    return out


def expected_groups(truth, layer, rank, world, key, group):
    """Group-wise metadata (scales / zero points, [K/g, N]): columns as above; rows follow K."""
    base = f"model.layers.{layer}."
    lg = truth["logical"]
    sets = col_sets(rank, world)
    out = {}
    out["qkv_proj"] = np.concatenate([lg[base + p][key][:, c] for p, c in sets["qkv"]], axis=1)
    out["gate_up_proj"] = np.concatenate([lg[base + p][key][:, c] for p, c in sets["gate_up"]], axis=1)
    for mod, proj in (("o_proj", "self_attn.o_proj"), ("down_proj", "mlp.down_proj")):
        full = lg[base + proj][key]
        k = full.shape[0] * group
        rows = row_set(k, rank, world)
        out[mod] = full[rows[0] // group:(rows[-1] + 1) // group, :]
    return out


def build(tmp, fmt, rank, world, dtype=torch.float16, kv_cache_dtype="auto", **kw):
    with simulated_tensor_parallel(rank, world):
        return L.load_model(str(tmp), dtype=dtype, kv_cache_dtype=kv_cache_dtype, device="cpu",
                            process_weights=False, **kw)


MODS = ("qkv_proj", "o_proj", "gate_up_proj", "down_proj")
WORLDS = [(0, 1), (0, 2), (1, 2), (0, 4), (3, 4)]


# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("embedded", [True, False])
def test_gptq_checkpoint(tmp_path, embedded):
    truth = CU.write_checkpoint(str(tmp_path), CFG, "gptq", seed=1, embedded_config=embedded)
    for rank, world in WORLDS:
        m = build(tmp_path, "gptq", rank, world, quantization=None if embedded else "gptq")
        for li, layer in enumerate(m.layers):
            eq = expected_logical(truth, li, rank, world, "q")
            es = expected_groups(truth, li, rank, world, "s", 128)
            ez = expected_groups(truth, li, rank, world, "zp", 128)
            for mod in MODS:
                lin = getattr(layer, mod)
                np.testing.assert_array_equal(oq.gptq_unpack(lin.qweight.numpy()), eq[mod], err_msg=f"{mod} q")
                np.testing.assert_array_equal(lin.scales.numpy(), es[mod], err_msg=f"{mod} scales")
                np.testing.assert_array_equal(oq.unpack_cols(lin.qzeros.numpy()), (ez[mod] - 1) & 15,
                                              err_msg=f"{mod} zeros (stored = zero - 1)")
                k_local = eq[mod].shape[0]
                k0 = row_set(k_local * world, rank, world)[0] if mod in ("o_proj", "down_proj") else 0
                np.testing.assert_array_equal(lin.g_idx.numpy(), (k0 + np.arange(k_local)) // 128)
        check_replicated(m, truth, rank, world)


@pytest.mark.parametrize("bits", [3, 8])
def test_gptq_3_and_8_bit_checkpoints_shard_like_4_bit(tmp_path, bits):
    """GPTQ 3-bit (32 values per 3 words: the pack factor is the Fraction 32/3, gptq.py:37) and 8-bit checkpoints through
    the same loader under TP 1 / 2 / 4: every rank's slice unpacks to the logical values."""
    truth = CU.write_checkpoint(str(tmp_path), CFG, "gptq", seed=4, bits=bits)
    mask = (1 << bits) - 1
    for rank, world in [(0, 1), (1, 2), (3, 4)]:
        m = build(tmp_path, "gptq", rank, world)
        for li, layer in enumerate(m.layers):
            eq = expected_logical(truth, li, rank, world, "q")
            ez = expected_groups(truth, li, rank, world, "zp", 128)
            for mod in MODS:
                lin = getattr(layer, mod)
                assert lin.quant_config.weight_bits == bits
                np.testing.assert_array_equal(oq.gptq_unpack(lin.qweight.numpy(), bits), eq[mod], err_msg=f"{mod} q")
                np.testing.assert_array_equal(oq.unpack_cols(lin.qzeros.numpy(), bits), (ez[mod] - 1) & mask,
                                              err_msg=f"{mod} zeros")


def check_replicated(m, truth, rank, world):
    t = {k: v.to(m.dtype) if v.dtype == torch.float16 else v for k, v in truth["tensors"].items()}
    assert torch.equal(m.embed_tokens.data, t["model.embed_tokens.weight"])
    assert torch.equal(m.norm.data, t["model.norm.weight"])
    rows = CFG.vocab_size // world
    assert torch.equal(m.lm_head.data, t["lm_head.weight"][rank * rows:(rank + 1) * rows])
    for li, layer in enumerate(m.layers):
        assert torch.equal(layer.input_layernorm.data, t[f"model.layers.{li}.input_layernorm.weight"])
        assert torch.equal(layer.post_attention_layernorm.data,
                           t[f"model.layers.{li}.post_attention_layernorm.weight"])


def test_awq_checkpoint(tmp_path):
    truth = CU.write_checkpoint(str(tmp_path), CFG, "awq", seed=2)
    for rank, world in WORLDS:
        m = build(tmp_path, "awq", rank, world)
        assert type(m.layers[0].qkv_proj.quant_method).__name__ == "CDNA4AWQLinearMethod"
        for li, layer in enumerate(m.layers):
            eq = expected_logical(truth, li, rank, world, "q")
            es = expected_groups(truth, li, rank, world, "s", 128)
            ez = expected_groups(truth, li, rank, world, "zp", 128)
            for mod in MODS:
                lin = getattr(layer, mod)
                np.testing.assert_array_equal(oq.awq_unpack(lin.qweight.numpy()), eq[mod], err_msg=mod)
                np.testing.assert_array_equal(oq.awq_unpack(lin.qzeros.numpy()), ez[mod], err_msg=mod)
                np.testing.assert_array_equal(lin.scales.numpy(), es[mod], err_msg=mod)


@pytest.mark.parametrize("fmt,static", [("fp8", False), ("fp8", True), ("ct-fp8-channel", False),
                                        ("ct-fp8-tensor", False), ("ct-fp8-tensor", True), ("ct-w8a16", False)])
def test_fp8_checkpoints(tmp_path, fmt, static):
    truth = CU.write_checkpoint(str(tmp_path), CFG, fmt, seed=3, fp8_static=static)
    lg, t = truth["logical"], truth["tensors"]
    for rank, world in WORLDS:
        m = build(tmp_path, fmt, rank, world, dtype=torch.bfloat16)
        sets = col_sets(rank, world)
        expect_method = {"fp8": "CDNA4Fp8LinearMethod", "ct-fp8-channel": "CompressedTensorsW8A8Fp8Method",
                         "ct-fp8-tensor": "CompressedTensorsW8A8Fp8Method",
                         "ct-w8a16": "CompressedTensorsW8A16Fp8Method"}[fmt]
        for li, layer in enumerate(m.layers):
            base = f"model.layers.{li}."
            for mod, group in (("qkv_proj", sets["qkv"]), ("gate_up_proj", sets["gate_up"])):
                lin = getattr(layer, mod)
                assert type(lin.quant_method).__name__ == expect_method
                w = torch.cat([lg[base + p]["wq"][torch.from_numpy(c)] for p, c in group], 0)
                assert torch.equal(lin.weight.data.view(torch.uint8), w.view(torch.uint8))
                if fmt in ("ct-fp8-channel", "ct-w8a16"):
                    s = torch.cat([lg[base + p]["s"][torch.from_numpy(c)] for p, c in group], 0)
                    assert torch.equal(lin.weight_scale.data, s)
                else:   # one scalar per logical matrix, in shard order
                    s = torch.stack([lg[base + p]["s"].reshape(()) for p, _ in group])
                    assert torch.equal(lin.weight_scale.data, s)
                if static:
                    i = torch.stack([t[base + p + ".input_scale"].reshape(()) for p, _ in group])
                    assert torch.equal(lin.input_scale.data, i)
                else:
                    assert getattr(lin, "input_scale", None) is None
            for mod, proj in (("o_proj", "self_attn.o_proj"), ("down_proj", "mlp.down_proj")):
                lin = getattr(layer, mod)
                full = lg[base + proj]["wq"]
                cols = torch.from_numpy(row_set(full.shape[1], rank, world))
                assert torch.equal(lin.weight.data.view(torch.uint8), full[:, cols].view(torch.uint8))
                assert torch.equal(lin.weight_scale.data.reshape(-1), lg[base + proj]["s"].reshape(-1))
        check_replicated(m, truth, rank, world)


@pytest.mark.parametrize("fmt,bits", [("ct-w4a16", 4), ("ct-w8a16i", 8)])
def test_compressed_tensors_pack_quantized(tmp_path, fmt, bits):
    """pack-quantized int4 (uint4b8) and int8 (uint8b128) weights (compressed_tensors_wNa16.py:16-20, 97-135)."""
    truth = CU.write_checkpoint(str(tmp_path), CFG, fmt, seed=4)
    for rank, world in WORLDS:
        m = build(tmp_path, fmt, rank, world)
        for li, layer in enumerate(m.layers):
            eq = expected_logical(truth, li, rank, world, "q")
            es = expected_groups(truth, li, rank, world, "s", 128)
            for mod in MODS:
                lin = getattr(layer, mod)
                assert type(lin.quant_method).__name__ == "CompressedTensorsWNA16Method"
                assert type(lin.kernel).__name__ == "CDNA4LinearKernel"
                # weight_packed is [N, K/8] ([N, K/4]) packed along K
                got = oq.gptq_unpack(np.ascontiguousarray(lin.weight_packed.numpy().T), bits)
                np.testing.assert_array_equal(got, eq[mod], err_msg=mod)
                np.testing.assert_array_equal(lin.weight_scale.numpy().T, es[mod], err_msg=mod)
                assert lin.kernel.config.weight_type.size_bits == bits


@pytest.mark.parametrize("fmt", ["gptq", "awq"])
def test_mixtral_int4_experts(tmp_path, fmt):
    """Mixtral-style checkpoint: router replicated, experts.{e}.w1/w3 column-cut into the two halves of
    the stacked w13 parameters, w2 row-cut (FusedMoE.weight_loader)."""
    cfg = M.TINY_MOE
    truth = CU.write_checkpoint(str(tmp_path), cfg, fmt, seed=21)
    lg, t = truth["logical"], truth["tensors"]
    inter = cfg.intermediate_size
    unpack = oq.gptq_unpack if fmt == "gptq" else oq.awq_unpack
    for rank, world in [(0, 1), (0, 2), (1, 2), (3, 4)]:
        m = build(tmp_path, fmt, rank, world)
        cols = np.arange(rank * inter // world, (rank + 1) * inter // world)
        for li, layer in enumerate(m.layers):
            assert layer.is_moe and layer.gate_up_proj is None
            assert torch.equal(layer.moe_gate.data, t[f"model.layers.{li}.block_sparse_moe.gate.weight"])
            ex = layer.experts
            for e in range(cfg.num_local_experts):
                base = f"model.layers.{li}.block_sparse_moe.experts.{e}."
                w1, w3, w2 = lg[base + "w1"], lg[base + "w3"], lg[base + "w2"]
                np.testing.assert_array_equal(unpack(ex.w13_qweight[e].numpy()),
                                              np.concatenate([w1["q"][:, cols], w3["q"][:, cols]], 1))
                np.testing.assert_array_equal(ex.w13_scales[e].numpy(),
                                              np.concatenate([w1["s"][:, cols], w3["s"][:, cols]], 1))
                np.testing.assert_array_equal(unpack(ex.w2_qweight[e].numpy()), w2["q"][cols, :])
                g0, g1 = cols[0] // 128, (cols[-1] + 1) // 128
                np.testing.assert_array_equal(ex.w2_scales[e].numpy(), w2["s"][g0:g1])
                z13 = np.concatenate([w1["zp"][:, cols], w3["zp"][:, cols]], 1)
                if fmt == "gptq":
                    np.testing.assert_array_equal(oq.unpack_cols(ex.w13_qzeros[e].numpy()), (z13 - 1) & 15)
                    np.testing.assert_array_equal(oq.unpack_cols(ex.w2_qzeros[e].numpy()),
                                                  (w2["zp"][g0:g1] - 1) & 15)
                    np.testing.assert_array_equal(ex.w2_g_idx[e].numpy(), cols // 128)
                else:
                    np.testing.assert_array_equal(oq.awq_unpack(ex.w13_qzeros[e].numpy()), z13)
                    np.testing.assert_array_equal(oq.awq_unpack(ex.w2_qzeros[e].numpy()), w2["zp"][g0:g1])
        # attention projections of the same checkpoint go through the dense plans
        eq = col_sets(rank, world)["qkv"]
        got = unpack(m.layers[0].qkv_proj.qweight.numpy())
        exp = np.concatenate([lg["model.layers.0." + p]["q"][:, c] for p, c in eq], 1)
        np.testing.assert_array_equal(got, exp)


def test_mixtral_compressed_tensors_experts(tmp_path):
    """llm-compressor W4A16 Mixtral checkpoint (pack-quantized, symmetric, group 128): experts.{e}.w{1,2,3}.weight_packed
    [N, K/8] / weight_scale [N, K/g] / weight_shape are transposed on their way into CompressedTensorsMoEMethod's
    parameters (is_transposed, fused_moe/layer.py:324-331) and cut like the GPTQ experts; the transposed weight_packed IS
    a GPTQ qweight."""
    cfg = M.TINY_MOE
    truth = CU.write_checkpoint(str(tmp_path), cfg, "ct-w4a16", seed=23)
    lg = truth["logical"]
    inter = cfg.intermediate_size
    for rank, world in [(0, 1), (1, 2), (3, 4)]:
        m = build(tmp_path, "ct-w4a16", rank, world)
        cols = np.arange(rank * inter // world, (rank + 1) * inter // world)
        for li, layer in enumerate(m.layers):
            ex = layer.experts
            assert type(ex.quant_method).__name__ == "CompressedTensorsMoEMethod"
            assert ex.w13_weight_packed.shape == (cfg.num_local_experts, cfg.hidden_size // 8, 2 * len(cols))
            for e in range(cfg.num_local_experts):
                base = f"model.layers.{li}.block_sparse_moe.experts.{e}."
                w1, w3, w2 = lg[base + "w1"], lg[base + "w3"], lg[base + "w2"]
                np.testing.assert_array_equal(oq.gptq_unpack(ex.w13_weight_packed[e].numpy()),
                                              np.concatenate([w1["q"][:, cols], w3["q"][:, cols]], 1))
                np.testing.assert_array_equal(ex.w13_weight_scale[e].numpy(),
                                              np.concatenate([w1["s"][:, cols], w3["s"][:, cols]], 1))
                np.testing.assert_array_equal(oq.gptq_unpack(ex.w2_weight_packed[e].numpy()), w2["q"][cols, :])
                g0, g1 = cols[0] // 128, (cols[-1] + 1) // 128
                np.testing.assert_array_equal(ex.w2_weight_scale[e].numpy(), w2["s"][g0:g1])
                assert ex.w13_weight_shape[e].tolist() == [inter, cfg.hidden_size]
                assert ex.w2_weight_shape[e].tolist() == [cfg.hidden_size, inter]


def test_compressed_tensors_experts_channel_strategy():
    """``channel`` strategy: ONE scale row per expert matrix; w1 / w3 rows are cut along N, w2's is NOT cut by tensor
    parallelism (fused_moe/layer.py:252-265)."""
    from aphrodite_engine_amd.moe import FusedMoE
    from aphrodite_engine_amd.quantization.compressed_tensors import CompressedTensorsConfig
    qc = CompressedTensorsConfig.from_config({"format": "pack-quantized", "config_groups": {"g": {
        "targets": ["Linear"], "weights": {"num_bits": 4, "type": "int", "symmetric": True, "strategy": "channel"},
        "input_activations": None}}})
    e, h, inter = 2, 256, 512
    rng = np.random.default_rng(0)
    for rank, world in [(0, 1), (1, 2)]:
        with simulated_tensor_parallel(rank, world):
            moe = FusedMoE(e, 2, h, inter, params_dtype=torch.float16, quant_config=qc, prefix="x.experts")
        assert moe.w13_weight_scale.shape == (e, 1, 2 * inter // world) and moe.w2_weight_scale.shape == (e, 1, h)
        cols = slice(rank * inter // world, (rank + 1) * inter // world)
        for x in range(e):
            for shard, n, k in (("w1", inter, h), ("w3", inter, h), ("w2", h, inter)):
                s = torch.from_numpy(rng.random((n, 1)).astype(np.float16))
                wp = torch.from_numpy(rng.integers(-2 ** 31, 2 ** 31 - 1, (n, k // 8)).astype(np.int32))
                fused = "w13_" if shard != "w2" else "w2_"
                moe.weight_loader(getattr(moe, fused + "weight_scale"), s, f"x.experts.{x}.{shard}.weight_scale", shard, x)
                moe.weight_loader(getattr(moe, fused + "weight_packed"), wp, f"x.experts.{x}.{shard}.weight_packed", shard, x)
                if shard == "w2":
                    assert torch.equal(moe.w2_weight_scale[x], s.t())
                    assert torch.equal(moe.w2_weight_packed[x], wp.t()[cols.start // 8:cols.stop // 8])
                else:
                    off = 0 if shard == "w1" else inter // world
                    assert torch.equal(moe.w13_weight_scale[x][:, off:off + inter // world], s.t()[:, cols])
                    assert torch.equal(moe.w13_weight_packed[x][:, off:off + inter // world], wp.t()[:, cols])


@pytest.mark.parametrize("fused", [False, True])
def test_unquantised_checkpoint_and_fused_on_disk(tmp_path, fused):
    truth = CU.write_checkpoint(str(tmp_path), CFG, "fp16", seed=5, fused_on_disk=fused)
    for rank, world in WORLDS:
        m = build(tmp_path, "fp16", rank, world)
        for li, layer in enumerate(m.layers):
            ew = expected_logical(truth, li, rank, world, "w")
            for mod in MODS:
                lin = getattr(layer, mod)
                assert lin.quant_method is None
                np.testing.assert_array_equal(lin.weight.numpy().T, ew[mod], err_msg=mod)


# ----------------------------------------------------------------------------------------------
def test_kv_cache_scales(tmp_path):
    n = CFG.num_hidden_layers
    CU.write_checkpoint(str(tmp_path / "a"), CFG, "fp8", seed=6, kv_scales="kv")
    m = build(tmp_path / "a", "fp8", 0, 1, kv_cache_dtype="fp8")
    assert [(l.k_scale, l.v_scale) for l in m.layers] == [
        (pytest.approx(0.02 + 0.001 * i), pytest.approx(0.03 + 0.001 * i)) for i in range(n)]
    # a 16-bit cache ignores them
    m = build(tmp_path / "a", "fp8", 0, 1, kv_cache_dtype="auto")
    assert all((l.k_scale, l.v_scale) == (1.0, 1.0) for l in m.layers)
    # deprecated single kv_scale: used for both
    CU.write_checkpoint(str(tmp_path / "b"), CFG, "fp8", seed=6, kv_scales="legacy")
    m = build(tmp_path / "b", "fp8", 0, 1, kv_cache_dtype="fp8")
    assert [(l.k_scale, l.v_scale) for l in m.layers] == [
        (pytest.approx(0.05 + 0.001 * i), pytest.approx(0.05 + 0.001 * i)) for i in range(n)]
    # compressed-tensors spelling
    CU.write_checkpoint(str(tmp_path / "c"), CFG, "ct-fp8-channel", seed=6, kv_scales="ct")
    m = build(tmp_path / "c", "ct", 0, 1, kv_cache_dtype="fp8")
    assert m.layers[1].k_scale == pytest.approx(0.021) and m.layers[1].v_scale == pytest.approx(0.031)
    # none in the checkpoint: 1.0
    CU.write_checkpoint(str(tmp_path / "d"), CFG, "fp8", seed=6)
    m = build(tmp_path / "d", "fp8", 0, 1, kv_cache_dtype="fp8")
    assert all((l.k_scale, l.v_scale) == (1.0, 1.0) for l in m.layers)


def test_quantization_param_path(tmp_path):
    CU.write_checkpoint(str(tmp_path / "m"), CFG, "fp8", seed=7)
    n = CFG.num_hidden_layers
    doc = {"model_type": "llama", "kv_cache": {"dtype": "float8_e4m3fn", "scaling_factor": {
        "0": {str(i): 0.1 + i for i in range(n)}, "1": {str(i): 0.2 + i for i in range(n)}}}}
    path = tmp_path / "kv.json"
    path.write_text(json.dumps(doc))
    m = build(tmp_path / "m", "fp8", 1, 2, kv_cache_dtype="fp8", quantization_param_path=str(path))
    assert [(l.k_scale, l.v_scale) for l in m.layers] == [(pytest.approx(0.2 + i), ) * 2 for i in range(n)]
    with pytest.raises(ValueError, match="TP size 2"):
        build(tmp_path / "m", "fp8", 0, 1, kv_cache_dtype="fp8", quantization_param_path=str(path))
    with pytest.raises(ValueError, match="fp8 KV cache"):
        build(tmp_path / "m", "fp8", 0, 2, kv_cache_dtype="auto", quantization_param_path=str(path))
    doc["kv_cache"]["dtype"] = "float8_e5m2"
    path.write_text(json.dumps(doc))
    with pytest.raises(ValueError, match="float8_e4m3fn"):
        build(tmp_path / "m", "fp8", 0, 2, kv_cache_dtype="fp8", quantization_param_path=str(path))
    doc["kv_cache"]["dtype"] = "float8_e4m3fn"
    del doc["kv_cache"]["scaling_factor"]["1"][str(n - 1)]
    path.write_text(json.dumps(doc))
    with pytest.raises(ValueError, match="malformed"):
        build(tmp_path / "m", "fp8", 1, 2, kv_cache_dtype="fp8", quantization_param_path=str(path))


def test_config_resolution_and_gates(tmp_path):
    CU.write_checkpoint(str(tmp_path / "g"), CFG, "gptq", seed=8)
    hf = L.read_hf_config(str(tmp_path / "g"))
    assert type(L.resolve_quant_config(str(tmp_path / "g"), hf)).__name__ == "GPTQConfig"
    with pytest.raises(ValueError, match="does not match"):
        L.resolve_quant_config(str(tmp_path / "g"), hf, quantization="awq")
    with pytest.raises(ValueError, match="not supported for quantization method"):
        L.resolve_quant_config(str(tmp_path / "g"), hf, dtype=torch.float32)
    # GPTQ without any config json
    CU.write_checkpoint(str(tmp_path / "n"), CFG, "fp16", seed=8)
    hf = L.read_hf_config(str(tmp_path / "n"))
    assert L.resolve_quant_config(str(tmp_path / "n"), hf) is None
    with pytest.raises(ValueError, match="Cannot find the config file"):
        L.resolve_quant_config(str(tmp_path / "n"), hf, quantization="gptq")
    with pytest.raises(ValueError, match="Invalid quantization method"):
        L.resolve_quant_config(str(tmp_path / "n"), hf, quantization="gguf")
    # schemes outside the hot path are refused, not mis-loaded
    from aphrodite_engine_amd.quantization.compressed_tensors import CompressedTensorsConfig
    int8 = CompressedTensorsConfig.from_config({"format": "int-quantized", "config_groups": {"g": {
        "targets": ["Linear"], "weights": {"num_bits": 8, "type": "int", "strategy": "channel"},
        "input_activations": {"num_bits": 8, "type": "int", "strategy": "token", "dynamic": True}}}})
    with pytest.raises(NotImplementedError):
        int8.get_quant_method(torch.nn.Module(), "model.layers.0.mlp.down_proj")
    mixed = CompressedTensorsConfig.from_config({"format": "float-quantized", "config_groups": {},
                                                 "ignore": ["model.layers.0.self_attn.q_proj"]})
    with pytest.raises(ValueError, match="same scheme"):
        mixed.get_quant_method(torch.nn.Module(), "model.layers.0.self_attn.qkv_proj")


def test_unsupported_architectures_are_refused(tmp_path):
    import json as _json
    CU.write_checkpoint(str(tmp_path), CFG, "fp16", seed=13)
    cfg_path = tmp_path / "config.json"
    base = _json.loads(cfg_path.read_text())
    for extra in ({"rope_scaling": {"rope_type": "longrope", "short_factor": [1.0], "long_factor": [2.0]}}, {"sliding_window": 4096},
                  {"hidden_act": "gelu"}):
        cfg_path.write_text(_json.dumps({**base, **extra}))
        with pytest.raises(NotImplementedError):
            build(tmp_path, "fp16", 0, 1)
    cfg_path.write_text(_json.dumps({**base, "sliding_window": None, "rope_scaling": None}))
    build(tmp_path, "fp16", 0, 1)
    # the scaled tables the reference's get_rope builds for Llama-architecture checkpoints load (rotary_embedding.py:940-973)
    for rs in ({"rope_type": "yarn", "factor": 4.0, "original_max_position_embeddings": 64}, {"type": "linear", "factor": 2.0},
               {"type": "dynamic", "factor": 2.0}):
        cfg_path.write_text(_json.dumps({**base, "rope_scaling": rs}))
        assert build(tmp_path, "fp16", 0, 1).cfg.rope_scaling == rs


@pytest.mark.parametrize("fmt", ["fp16", "gptq"])
@pytest.mark.parametrize("rank,world", WORLDS)
def test_projection_biases_are_sharded_like_their_layers(tmp_path, fmt, rank, world):
    """config.attention_bias / mlp_bias (models/llama.py:62-82, 135-150, 206-211): q / k / v and gate / up biases are cut along
    the output with their heads / columns (ColumnParallelLinear, linear.py:283-291), o / down biases are whole on every rank
    (RowParallelLinear adds them after the all-reduce, linear.py:1136-1150); such a layer stays off the fused steps."""
    truth = CU.write_checkpoint(str(tmp_path), CFG, fmt, seed=23, proj_bias="both")
    m = build(tmp_path, fmt, rank, world)
    hd = CFG.hidden_size // CFG.num_attention_heads
    hq, hkv, inter = CFG.num_attention_heads // world, max(1, CFG.num_key_value_heads // world), CFG.intermediate_size // world
    kv_rank = rank * CFG.num_key_value_heads // world if CFG.num_key_value_heads >= world else rank // (world // CFG.num_key_value_heads)
    for li, layer in enumerate(m.layers):
        lg = lambda p: torch.from_numpy(truth["logical"][f"model.layers.{li}.{p}"]["bias"])
        assert layer.has_bias and not layer.fused_decode_ok(4)
        want_qkv = torch.cat([lg("self_attn.q_proj")[rank * hq * hd:(rank + 1) * hq * hd],
                              lg("self_attn.k_proj")[kv_rank * hd:(kv_rank + hkv) * hd],
                              lg("self_attn.v_proj")[kv_rank * hd:(kv_rank + hkv) * hd]])
        assert torch.equal(layer.qkv_proj.bias.data, want_qkv)
        assert torch.equal(layer.gate_up_proj.bias.data, torch.cat([lg("mlp.gate_proj")[rank * inter:(rank + 1) * inter],
                                                                    lg("mlp.up_proj")[rank * inter:(rank + 1) * inter]]))
        assert torch.equal(layer.o_proj.bias.data, lg("self_attn.o_proj"))
        assert torch.equal(layer.down_proj.bias.data, lg("mlp.down_proj"))


def test_gptq_act_order_checkpoint(tmp_path):
    """desc_act = true: g_idx is a real permutation-derived map; column-parallel layers take it whole,
    row-parallel layers their K slice, and with TP > 1 the row-parallel group metadata is NOT cut
    (gptq.py:133-143: exllama is disabled there and every group must be reachable)."""
    import json as _json
    truth = CU.write_checkpoint(str(tmp_path), CFG, "gptq", seed=14)
    t = truth["tensors"]
    # turn the checkpoint into an act-order one: shuffle every g_idx (the loader must carry it through)
    from safetensors.torch import load_file, save_file
    rng = np.random.default_rng(0)
    for f in sorted(tmp_path.glob("*.safetensors")):
        d = load_file(str(f))
        for name in list(d):
            if name.endswith(".g_idx"):
                k = d[name].numel()
                d[name] = torch.from_numpy((rng.permutation(k) // 128).astype(np.int32))
                t[name] = d[name]
        save_file(d, str(f))
    cfgp = tmp_path / "config.json"
    c = _json.loads(cfgp.read_text())
    c["quantization_config"]["desc_act"] = True
    cfgp.write_text(_json.dumps(c))
    for rank, world in [(0, 1), (1, 2)]:
        m = build(tmp_path, "gptq", rank, world)
        for li, layer in enumerate(m.layers):
            base = f"model.layers.{li}."
            assert torch.equal(layer.qkv_proj.g_idx.data, t[base + "self_attn.v_proj.g_idx"])   # last shard written
            k = t[base + "mlp.down_proj.g_idx"].numel() // world
            assert torch.equal(layer.down_proj.g_idx.data, t[base + "mlp.down_proj.g_idx"][rank * k:(rank + 1) * k])
            full_groups = t[base + "mlp.down_proj.scales"].shape[0]
            assert layer.down_proj.scales.shape[0] == full_groups          # all groups on every rank
            assert torch.equal(layer.down_proj.scales.data, t[base + "mlp.down_proj.scales"])
            from aphrodite_engine_amd.quantization.gptq import ExllamaState
            assert layer.down_proj.exllama_state == (ExllamaState.UNUSED if world > 1 else ExllamaState.UNINITIALIZED)


def test_shard_plan_rules():
    p = L.qkv_plan(32, 8, 128, rank=5, world=16)            # 2 ranks per KV head
    assert [(o.local, o.src_index) for o in p.outs] == [(256, 5), (128, 2), (128, 2)]
    p = L.qkv_plan(32, 8, 128, rank=3, world=4)
    assert [(o.local, o.src_index) for o in p.outs] == [(1024, 3), (256, 3), (256, 3)]
    with pytest.raises(ValueError):
        L.qkv_plan(32, 8, 128, rank=0, world=3)
    with pytest.raises(ValueError):
        L.qkv_plan(32, 8, 128, rank=0, world=12)
    with pytest.raises(ValueError):
        L.merged_plan([14336, 14336], rank=0, world=3)
    # shape mismatches are errors, never silent truncation
    plan = L.merged_plan([16, 16], rank=0, world=1)
    from aphrodite_engine_amd.quantization.base_config import _param
    prm = _param(torch.zeros(32, 8), input_dim=1, output_dim=0)
    with pytest.raises(ValueError, match="does not fit"):
        L.load_sharded(plan, prm, torch.zeros(16, 9), 0)
    with pytest.raises(ValueError, match="Unknown shard id"):
        L.load_sharded(plan, prm, torch.zeros(16, 8), "q")
    with pytest.raises(KeyError):
        L.map_llama_name("model.layers.0.self_attn.unknown_proj.weight")


# ----------------------------------------------------------------------------------------------
def _tp_load_worker(rank, world, port, path, q):
    import torch.distributed as dist
    from aphrodite_engine_amd import distributed as d
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    d.init_tensor_parallel(world, backend="gloo")
    m = L.load_model(path, device="cpu", process_weights=False)
    lin = m.layers[1].qkv_proj
    # the ranks' lm_head shards gathered over the TP group give back the full matrix
    full = d.tensor_model_parallel_all_gather(m.lm_head.data.float(), dim=0)
    q.put((rank, oq.gptq_unpack(lin.qweight.numpy()).tolist(), full.shape[0], float(full.sum())))
    dist.destroy_process_group()


def test_load_under_real_tensor_parallel_gloo(tmp_path):
    """world_size 2 over gloo: each rank loads its own shard from the same directory."""
    truth = CU.write_checkpoint(str(tmp_path), CFG, "gptq", seed=9)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_tp_load_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in range(2)), key=lambda r: r[0])
    for p in procs:
        p.join(60)
    for rank, qkv, rows, total in res:
        exp = expected_logical(truth, 1, rank, 2, "q")["qkv_proj"]
        np.testing.assert_array_equal(np.array(qkv), exp)
        assert rows == CFG.vocab_size
        assert total == pytest.approx(float(truth["tensors"]["lm_head.weight"].float().sum()), rel=1e-6)


# ----------------------------------------------------------------------------------------------
def test_shard_arithmetic_matches_the_reference_weight_loaders(golden_dir):
    """tests/golden/loader_shards.npz holds what the reference's OWN MergedColumnParallelLinear /
    QKVParallelLinear / RowParallelLinear.weight_loader methods (lifted out of linear.py and run in the
    build container) leave in GPTQ / AWQ / FP8 / compressed-tensors parameters for (world, rank) =
    (1,0), (2,1), (4,3): the single copy routine of loader.py must produce the same bytes."""
    from aphrodite_engine_amd.quantization.base_config import _param
    g = np.load(os.path.join(golden_dir, "loader_shards.npz"))
    HQ, HKV, HD, INTER = 4, 2, 16, 96
    row_attrs = {"gptq.qzeros": dict(input_dim=0), "gptq.scales": dict(input_dim=0)}
    attrs = {
        "gptq.qweight": dict(input_dim=0, output_dim=1, packed_dim=0, pack_factor=8),
        "gptq.qzeros": dict(output_dim=1, packed_dim=1, pack_factor=8),
        "gptq.scales": dict(output_dim=1),
        "gptq.g_idx": dict(input_dim=0),
        "awq.qweight": dict(input_dim=0, output_dim=1, packed_dim=1, pack_factor=8),
        "fp8.weight": dict(input_dim=1, output_dim=0),
        "fp8.weight_scale": dict(needs_scalar_to_array=True),
        "ct.weight_scale": dict(output_dim=0),
    }
    shard_ids = {"qkv": ["q", "k", "v"], "merged": [0, 1], "row": [None]}
    checked = 0
    for key in g.files:
        if not key.startswith("out."):
            continue
        _, lname, fmt, pname, tag = key.split(".")
        name = f"{fmt}.{pname}"
        world, rank = int(tag[1:tag.index("r")]), int(tag[tag.index("r") + 1:])
        if lname.startswith("qkv_h"):             # extra head layouts: qkv_h8kv8, qkv_h4kv1, qkv_h8kv2
            hq, hkv = (int(v) for v in lname[len("qkv_h"):].split("kv"))
            plan, sids = L.qkv_plan(hq, hkv, HD, rank, world), shard_ids["qkv"]
        else:
            plan = {"qkv": lambda: L.qkv_plan(HQ, HKV, HD, rank, world),
                    "merged": lambda: L.merged_plan([INTER, INTER], rank, world),
                    "row": lambda: L.row_plan(rank, world)}[lname]()
            sids = shard_ids[lname]
        want = torch.from_numpy(g[key])
        a = dict(attrs[name])
        if lname == "row":
            a.update(row_attrs.get(name, {}))
        prm = _param(torch.zeros_like(want), **a)
        for sid in sids:
            L.load_sharded(plan, prm, torch.from_numpy(g[f"in.{lname}.{name}.{sid}"]), sid)
        assert torch.equal(prm.data, want), key
        # fused on disk: one tensor, shard id None
        if lname != "row" and "output_dim" in a:
            fused = torch.cat([torch.from_numpy(g[f"in.{lname}.{name}.{sid}"]) for sid in sids], dim=a["output_dim"])
            prm2 = _param(torch.zeros_like(want), **a)
            L.load_sharded(plan, prm2, fused, None)
            assert torch.equal(prm2.data, want), key + " (fused)"
        checked += 1
    assert checked >= 90


def test_vocab_not_divisible_by_tp_is_padded_not_truncated(tmp_path):
    """A checkpoint with added tokens (vocab 1000: not a multiple of 64, nor of tp 4): the lm_head is the vocabulary
    padded to 1024 and sharded like ParallelLMHead (vocab_parallel_embedding.py); the last rank holds the 232 real tail
    rows + 24 zero rows -- no token is dropped."""
    import dataclasses
    from aphrodite_engine_amd.model import padded_vocab_size
    cfg = dataclasses.replace(CFG, vocab_size=1000)
    truth = CU.write_checkpoint(str(tmp_path), cfg, "fp16", seed=9)
    full = truth["tensors"]["lm_head.weight"]
    assert padded_vocab_size(1000, 4) == 1024 and padded_vocab_size(128256, 8) == 128256 and padded_vocab_size(1000, 3) == 1152
    seen = []
    for rank in range(4):
        m = build(tmp_path, "fp16", rank, 4)
        assert m.lm_head.shape[0] == 256 and m.vocab_padded == 1024
        lo = rank * 256
        valid = max(0, min(256, 1000 - lo))
        assert torch.equal(m.lm_head.data[:valid], full[lo:lo + valid])
        assert not m.lm_head.data[valid:].any()
        seen.append(m.lm_head.data[:valid])
    assert torch.equal(torch.cat(seen), full)
    m1 = build(tmp_path, "fp16", 0, 1)
    assert m1.lm_head.shape[0] == 1024 and torch.equal(m1.lm_head.data[:1000], full)
    hidden = torch.randn(3, cfg.hidden_size).half()
    assert m1.compute_logits(hidden).shape == (3, 1000)
