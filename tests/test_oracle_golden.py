"""Pins oracle/ against golden vectors produced by the reference's own Python
(tests/golden/make_golden.py) and against the reference's own CPU kernels
(oracle/_ref, compiled from /root/reference/kernels/cpu)."""
import os

import numpy as np
import pytest
import torch

from oracle import attention as oa
from oracle import fp8 as of8
from oracle import quant as oq


@pytest.fixture(scope="module")
def int4(golden_dir):
    return np.load(os.path.join(golden_dir, "int4_formats.npz"))


@pytest.fixture(scope="module")
def f8(golden_dir):
    return np.load(os.path.join(golden_dir, "fp8.npz"))


@pytest.fixture(scope="module")
def att(golden_dir):
    return np.load(os.path.join(golden_dir, "attention.npz"))


# ---- int4 formats -----------------------------------------------------------
def test_quantize_weights_symmetric(int4):
    w_ref, w_q, w_s, zp = oq.quantize_weights(int4["w"], 4, 128)
    assert zp is None
    np.testing.assert_array_equal(w_q, int4["sym_w_q"])
    np.testing.assert_array_equal(w_s, int4["sym_w_s"])
    np.testing.assert_array_equal(w_ref, int4["sym_w_ref"])


def test_quantize_weights_zero_points(int4):
    w_ref, w_q, w_s, zp = oq.quantize_weights(int4["w"], 4, 128, zero_points=True)
    np.testing.assert_array_equal(w_q, int4["zp_w_q"])
    np.testing.assert_array_equal(zp, int4["zp_w_zp"])
    np.testing.assert_array_equal(w_s, int4["zp_w_s"])
    np.testing.assert_array_equal(w_ref, int4["zp_w_ref"])


def test_gptq_pack_unpack(int4):
    packed = oq.gptq_pack(int4["sym_w_q"])
    np.testing.assert_array_equal(packed, int4["gptq_packed"])
    np.testing.assert_array_equal(oq.gptq_unpack(packed), int4["sym_w_q"])


def test_awq_pack_unpack(int4):
    np.testing.assert_array_equal(oq.awq_pack(int4["zp_w_q"]), int4["awq_packed"])
    np.testing.assert_array_equal(oq.awq_pack(int4["zp_w_zp"]),
                                  int4["awq_zeros_packed"])
    np.testing.assert_array_equal(oq.awq_unpack(int4["awq_packed"]),
                                  int4["zp_w_q"])
    np.testing.assert_array_equal(oq.pack_cols(int4["zp_w_q"]), int4["pack_cols"])
    np.testing.assert_array_equal(oq.unpack_cols(int4["pack_cols"]),
                                  int4["unpack_cols"])


def test_awq_dequantize_exact(int4):
    """Reference test pins AWQ dequant exactly (test_awq_triton.py:96)."""
    got = oq.awq_dequantize(int4["awq_packed"], int4["zp_w_s"],
                            int4["awq_zeros_packed"])
    np.testing.assert_array_equal(got, int4["awq_dequant"])
    # and it reproduces quantize_weights' own w_ref
    np.testing.assert_array_equal(got, int4["zp_w_ref"])


def test_act_order_sort(int4):
    """sort_weights (quant_utils.py:313-331): argsort(g_idx) gathers rows."""
    perm = np.argsort(int4["act_g_idx"], kind="stable")
    # torch.argsort is not guaranteed stable: compare the sorted results
    np.testing.assert_array_equal(int4["act_g_idx"][int4["act_sort_idx"]],
                                  int4["act_sorted_g"])
    np.testing.assert_array_equal(int4["act_w_q"][int4["act_sort_idx"]],
                                  int4["act_sorted_q"])
    np.testing.assert_array_equal(np.sort(int4["act_g_idx"]), int4["act_g_idx"][perm])


def test_gptq_shuffle_consistency(int4):
    """Self-consistency of the restatement (the pin against the reference's own kernels is
    test_gptq_*_reference_kernels below): shuffled+perm
    dequant followed by the A-gather equals the plain act-order dequant."""
    rng = np.random.RandomState(0)
    q = int4["act_w_q"]                        # rows permuted by rand_perm
    g_idx = int4["act_g_idx"]
    scales = int4["sym_w_s"]
    G, N = scales.shape
    zeros = np.full((G, N), 8, np.int32)       # uint4b8: zero point 8
    qzeros = oq.gptq_pack_zeros(zeros)
    qweight = oq.gptq_pack(q)
    a = rng.randn(5, q.shape[0]).astype(np.float16)
    ref = oq.gptq_gemm(a, qweight, qzeros, scales, g_idx, use_exllama=False)
    w = oq.gptq_dequant(qweight, qzeros, scales, g_idx)
    # the reference rounds (q - 8) * s to fp16 (quant_utils.py:184)
    np.testing.assert_array_equal(w.astype(np.float16), int4["act_w_ref"])
    perm = np.argsort(g_idx, kind="stable").astype(np.int32)
    shuf = oq.gptq_shuffle(qweight, perm)
    got = oq.gptq_gemm(a, shuf, qzeros, scales, perm, use_exllama=True)
    np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-9)
    # nibble permutation is an involution pair
    w32 = qweight.view(np.uint32)
    np.testing.assert_array_equal(
        oq.unshuffle_4bit_word(oq.shuffle_4bit_word(w32)), w32)
    # known answer: element i in nibble i -> 0x7531_6420
    assert int(oq.shuffle_4bit_word(np.array([0x76543210], np.uint32))[0]) == 0x75316420


# ---- fp8 -----------------------------------------------------------------
def test_fp8_tables(f8):
    for kind in ("e4m3", "e5m2"):
        mine = of8.fp8_decode_table(kind)
        ref = f8[f"{kind}_table"]
        np.testing.assert_array_equal(np.isnan(mine), np.isnan(ref))
        ok = ~np.isnan(ref)
        np.testing.assert_array_equal(mine[ok], ref[ok])
        fin = ok & np.isfinite(ref)
        bits = np.arange(256, dtype=np.uint8)[fin]
        enc = of8.fp8_encode(ref[fin], kind)
        # -0 and +0 both decode to 0; compare decoded values
        np.testing.assert_array_equal(of8.fp8_decode(enc, kind), ref[fin])
        assert (enc == bits).mean() > 0.99


def test_fp8_saturation():
    assert of8.fp8_decode(of8.fp8_encode([1e6, -1e6], "e4m3"), "e4m3").tolist() == [448.0, -448.0]
    assert of8.fp8_decode(of8.fp8_encode([1e9, -1e9], "e5m2"), "e5m2").tolist() == [57344.0, -57344.0]


def test_fp8_per_token_quant(f8):
    q, s = of8.dynamic_per_token_scaled_fp8_quant(f8["x"])
    np.testing.assert_array_equal(s, f8["s_tok"])
    np.testing.assert_array_equal(q, f8["q_tok"])
    q, s = of8.dynamic_per_token_scaled_fp8_quant(f8["x"], f8["ub"][0])
    np.testing.assert_array_equal(s, f8["s_ub"])
    np.testing.assert_array_equal(q, f8["q_ub"])


def test_fp8_per_tensor_quant(f8):
    q, s = of8.dynamic_scaled_fp8_quant(f8["x"])
    np.testing.assert_array_equal(s, f8["s_ten"])
    np.testing.assert_array_equal(q, f8["q_ten"])


def test_scaled_mm(f8):
    got = of8.scaled_mm(f8["mm_a"], f8["mm_b"], f8["mm_sa"], f8["mm_sb"],
                        f8["mm_bias"])
    np.testing.assert_allclose(got, f8["mm_out"], rtol=1e-5, atol=1e-4)


# ---- attention / cache -------------------------------------------------------
@pytest.mark.parametrize("tag", ["plain", "alibi"])
def test_paged_attention_golden(att, tag):
    slopes = att["slopes"] if tag == "alibi" else None
    got = oa.paged_attention_decode(att["q"], att["kc"], att["vc"], att["bt"],
                                    att["seq_lens"], 0.125, slopes)
    np.testing.assert_allclose(got, att["out_" + tag], rtol=1e-4, atol=2e-5)
    merged, mx, es, tmp = oa.paged_attention_v2_partials(
        att["q"], att["kc"], att["vc"], att["bt"], att["seq_lens"], 0.125,
        partition_size=64, alibi_slopes=slopes)
    np.testing.assert_allclose(merged, got, rtol=1e-4, atol=1e-5)


def test_reshape_and_cache_golden(att):
    kc, vc = att["kc"].copy(), att["vc"].copy()
    oa.reshape_and_cache(att["rc_key"], att["rc_val"], kc, vc, att["rc_slots"])
    np.testing.assert_array_equal(kc, att["rc_kc"])
    np.testing.assert_array_equal(vc, att["rc_vc"])


def _ref_lib():
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                        "oracle", "_ref", "libaphro_ref_cpu.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    try:
        torch.ops.load_library(path)
    except Exception as e:  # pragma: no cover
        pytest.skip(f"cannot load oracle/_ref: {e}")
    return torch.ops.aphro_ref_cpu


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("version", ["v1", "v2"])
def test_oracle_vs_reference_cpu_kernel(dtype, version):
    """Our restatement vs the reference's own compiled CPU paged attention
    (kernels/cpu/attention.cpp) on seeded inputs."""
    ops = _ref_lib()
    torch.manual_seed(0)
    S, Hq, Hkv, D, BS, NB = 5, 8, 2, 128, 16, 80
    x = 16 // torch.tensor([], dtype=dtype).element_size()
    q = (torch.randn(S, Hq, D) * 0.3).to(dtype)
    kc = ((torch.rand(NB, Hkv, D // x, BS, x) - 0.5) * 0.4).to(dtype)
    vc = ((torch.rand(NB, Hkv, D, BS) - 0.5) * 0.4).to(dtype)
    seq_lens = torch.tensor([1, 15, 16, 700, 1100], dtype=torch.int32)
    maxb = (1100 + BS - 1) // BS
    bt = torch.stack([torch.randperm(NB)[:maxb] for _ in range(S)]).int()
    out = torch.empty_like(q)
    scale = D ** -0.5
    if version == "v1":
        ops.paged_attention_v1(out, q, kc, vc, Hkv, scale, bt, seq_lens, BS,
                               1100, None, "auto", 1.0, 1.0, 0, 0, 0, 64, 0)
    else:
        parts = (1100 + 511) // 512
        es = torch.empty(S, Hq, parts)
        ml = torch.empty(S, Hq, parts)
        tmp = torch.empty(S, Hq, parts, D, dtype=dtype)
        ops.paged_attention_v2(out, es, ml, tmp, q, kc, vc, Hkv, scale, bt,
                               seq_lens, BS, 1100, None, "auto", 1.0, 1.0, 0, 0,
                               0, 64, 0)
    ref = oa.paged_attention_decode(q, kc.float().numpy(), vc.float().numpy(),
                                    bt.numpy(), seq_lens.numpy(), scale)
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    np.testing.assert_allclose(out.float().numpy(), ref, rtol=tol, atol=tol)


def test_reshape_and_cache_vs_reference_cpu_kernel():
    ops = _ref_lib()
    torch.manual_seed(1)
    T, Hkv, D, BS, NB = 11, 2, 64, 16, 6
    x = 4
    key = torch.randn(T, Hkv, D)
    val = torch.randn(T, Hkv, D)
    kc = torch.zeros(NB, Hkv, D // x, BS, x)
    vc = torch.zeros(NB, Hkv, D, BS)
    slots = torch.randperm(NB * BS)[:T].to(torch.int64)
    kc2, vc2 = kc.numpy().copy(), vc.numpy().copy()
    ops.reshape_and_cache(key, val, kc, vc, slots, "auto", 1.0, 1.0)
    oa.reshape_and_cache(key, val, kc2, vc2, slots.numpy())
    np.testing.assert_array_equal(kc.numpy(), kc2)
    np.testing.assert_array_equal(vc.numpy(), vc2)


def test_moe_routing_oracle_known_answers():
    """moe_align_block_size: the worked example in the reference's own docstring
    (modeling/layers/fused_moe/fused_moe.py:199-212; experts 1..4 there are 0..3 here);
    topk_softmax: torch softmax + topk (tests/kernels/test_moe.py:19-26)."""
    import torch
    from oracle import moe
    ids = np.array([[2, 3, 4], [1, 2, 4], [1, 3, 4], [1, 2, 3]]) - 1
    sorted_ids, expert_ids, post = moe.moe_align_block_size(ids, 4, 4)
    assert sorted_ids[:16].tolist() == [3, 6, 9, 12, 0, 4, 10, 12, 1, 7, 11, 12, 2, 5, 8, 12]
    assert expert_ids[:4].tolist() == [0, 1, 2, 3] and post == 16
    rng = np.random.default_rng(0)
    g = rng.standard_normal((9, 8)).astype(np.float32)
    w, idx, src = moe.topk_softmax(g, 2)
    tw, ti = torch.topk(torch.softmax(torch.from_numpy(g), dim=-1), 2, dim=-1)
    np.testing.assert_array_equal(idx, ti.numpy())
    np.testing.assert_allclose(w, tw.numpy(), rtol=1e-6)
    assert src[3, 1] == 1 * 9 + 3



# ---------------------------------------------------------------------------
# sampler restatement vs the reference's own functions (tests/golden/sampler.npz)
# ---------------------------------------------------------------------------
def test_sampler_golden(golden_dir):
    from oracle import sampling as osamp
    g = np.load(os.path.join(golden_dir, "sampler.npz"))
    logits, top_k, top_p = g["logits"], g["top_k"], g["top_p"]
    masked = osamp.apply_top_k_top_p(logits, top_p, top_k)
    tie_row = 6
    for r in range(logits.shape[0]):
        if r == tie_row:
            # torch.sort and numpy's stable sort order equal values differently: same kept VALUES
            np.testing.assert_array_equal(np.sort(masked[r][np.isfinite(masked[r])]),
                                          np.sort(g["masked"][r][np.isfinite(g["masked"][r])]))
        else:
            np.testing.assert_array_equal(masked[r], g["masked"][r], err_msg=f"row {r}")
    np.testing.assert_allclose(osamp.softmax32(g["masked"]), g["probs"], rtol=2e-6, atol=1e-12)
    # min-p on the reference's masked logits
    np.testing.assert_array_equal(osamp.apply_min_p(g["masked"], g["min_p"]), g["masked_minp"])
    # the draw itself, on the reference's probabilities and its Exp(1) variates
    np.testing.assert_array_equal(osamp.multinomial(g["probs"], g["q"]), g["ids"])
    # and the whole chain from logits
    ids, _ = osamp.sample(logits, np.ones(logits.shape[0], np.float32), top_k, top_p, g["q"])
    keep = np.arange(logits.shape[0]) != tie_row
    np.testing.assert_array_equal(ids[keep], g["ids"][keep])


def test_compressed_tensors_ignore_list_golden(golden_dir):
    """Host logic, not oracle: layer_is_ignored vs the reference's should_ignore_layer."""
    import json
    from aphrodite_engine_amd.quantization.utils import layer_is_ignored
    cases = json.load(open(os.path.join(golden_dir, "ct_ignore.json")))
    assert len(cases) >= 40
    for c in cases:
        if c["result"] == "ValueError":
            with pytest.raises(ValueError):
                layer_is_ignored(c["layer"], c["ignore"])
        else:
            assert layer_is_ignored(c["layer"], c["ignore"]) == c["result"], c


@pytest.mark.parametrize("tag,window", [("full", 0), ("win8", 8)])
def test_prefill_over_paged_cache_golden(golden_dir, tag, window):
    """oracle.attention.context_attention / varlen_causal_attention vs the reference's ref_paged_attn (flash-attn
    tests): cached context + causal new tokens, GQA, shuffled block tables, sliding window."""
    g = np.load(os.path.join(golden_dir, "prefill_paged.npz"))
    q, kc, vc, bt = g["q"], g["kc"], g["vc"], g["bt"]
    q_lens, kv_lens, scale = g["q_lens"], g["kv_lens"], float(g["scale"])
    nb, bs, hkv, hd = kc.shape
    # logical K/V of every sequence from the flash-layout cache [blocks, block, heads, dim]
    seqs_k = [kc[bt[i]].reshape(-1, hkv, hd)[:kv_lens[i]] for i in range(len(q_lens))]
    seqs_v = [vc[bt[i]].reshape(-1, hkv, hd)[:kv_lens[i]] for i in range(len(q_lens))]
    k_new = np.concatenate([seqs_k[i][kv_lens[i] - q_lens[i]:] for i in range(len(q_lens))], 0)
    v_new = np.concatenate([seqs_v[i][kv_lens[i] - q_lens[i]:] for i in range(len(q_lens))], 0)
    # the same context in the paged layout of the decode kernels: K [blocks, heads, dim/x, block, x], V [blocks, heads, dim, block]
    x = 4      # float32 cache
    pk = np.ascontiguousarray(kc.transpose(0, 2, 3, 1).reshape(nb, hkv, hd // x, x, bs).transpose(0, 1, 2, 4, 3))
    pv = np.ascontiguousarray(vc.transpose(0, 2, 3, 1))
    start = np.concatenate([[0], np.cumsum(q_lens)])[:-1]
    out = oa.context_attention(q, k_new, v_new, pk, pv, bt, start, kv_lens, kv_lens - q_lens, scale,
                               sliding_window=window)
    np.testing.assert_allclose(out, g["out_" + tag], rtol=2e-5, atol=2e-6)
    if window == 0:   # sequences without cached context are plain causal prefill
        sel = [i for i in range(len(q_lens)) if q_lens[i] == kv_lens[i]]
        rows = np.concatenate([np.arange(start[i], start[i] + q_lens[i]) for i in sel])
        cu = np.concatenate([[0], np.cumsum([q_lens[i] for i in sel])])
        out2 = oa.varlen_causal_attention(q[rows], k_new[rows], v_new[rows], cu, scale)
        np.testing.assert_allclose(out2, g["out_full"][rows], rtol=2e-5, atol=2e-6)


def test_moe_layer_golden(golden_dir):
    """oracle/moe.py (routing + dense per-expert MLP) vs the reference's torch_moe."""
    from oracle import moe as om
    g = np.load(os.path.join(golden_dir, "moe.npz"))
    w, ids, _ = om.topk_softmax(g["score"], int(g["topk"]))
    # torch_moe: softmax -> topk, weights NOT renormalised; w1 is [E, 2N, K] (gate rows | up rows)
    out = om.moe_layer(g["a"], np.transpose(g["w1"], (0, 2, 1)), np.transpose(g["w2"], (0, 2, 1)), w, ids)
    np.testing.assert_allclose(out, g["out"], rtol=2e-4, atol=2e-5)


def test_fp8_requantize_with_max_scale_golden(golden_dir):
    """Host logic: fused shards quantised with their own scales are requantised to the largest one exactly as
    the reference does it (through fp16, multiplying by 1 / scale) -- same bytes."""
    from aphrodite_engine_amd.quantization.fp8 import requantize_with_max_scale
    g = np.load(os.path.join(golden_dir, "fp8_requant.npz"))
    w = torch.from_numpy(g["w"]).view(torch.float8_e4m3fn)
    mx, out = requantize_with_max_scale(w.clone(), torch.from_numpy(g["scales"]), [int(x) for x in g["widths"]])
    assert float(mx) == float(g["max_scale"])
    np.testing.assert_array_equal(out.view(torch.uint8).numpy(), g["out"])
    assert (g["out"] != g["w"]).any()
    # a module stored fused on disk carries one real scale: nothing is requantised
    mx2, out2 = requantize_with_max_scale(w.clone(), torch.from_numpy(g["fused_scales"]), [int(x) for x in g["widths"]])
    assert float(mx2) == float(g["max_scale_fused"])
    np.testing.assert_array_equal(out2.view(torch.uint8).numpy(), g["out_fused"])
    np.testing.assert_array_equal(g["out_fused"], g["w"])


def test_expert_params_mapping_golden(golden_dir):
    import json
    from aphrodite_engine_amd.moe import FusedMoE
    want = json.load(open(os.path.join(golden_dir, "moe_mapping.json")))
    got = [list(r) for r in FusedMoE.make_expert_params_mapping("w1", "w2", "w3", 3)] + \
        [list(r) for r in FusedMoE.make_expert_params_mapping("gate_proj", "down_proj", "up_proj", 2)]
    assert got == want


def test_kv_scale_rules_golden(golden_dir):
    """Host logic vs the reference: BaseKVCacheMethod.process_weights_after_loading (checkpoint scales) and
    kv_cache_scales_loader (quantization_param_path json)."""
    import json
    import tempfile
    from aphrodite_engine_amd import loader as L
    doc = json.load(open(os.path.join(golden_dir, "kv_scales.json")))
    for c in doc["rule"]:
        found = {0: {}}
        if c["k_scale"] > 0:
            found[0]["k"] = c["k_scale"]
        if c["v_scale"] > 0:
            found[0]["v"] = c["v_scale"]
        k, v = L.finalize_kv_scales(found, 1, c["kv_cache_dtype"])[0]
        want = c["out"] if c["out"][0] is not None else [1.0, 1.0]      # "auto": the reference leaves the default 1.0
        assert (k, v) == (pytest.approx(want[0], rel=1e-6), pytest.approx(want[1], rel=1e-6)), c
    for c in doc["param_path"]:
        with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as tf:
            json.dump(c["doc"], tf)
        a = c["args"]
        try:
            if c["result"]:
                got = L.read_kv_cache_scales(tf.name, a["tp_rank"], a["tp_size"], a["layers"], a["model_type"])
                assert sorted(got.items()) == [(i, pytest.approx(s)) for i, s in c["result"]]
            else:
                # the reference logs and silently falls back to 1.0 for every layer; here a malformed file is an error
                with pytest.raises(ValueError):
                    L.read_kv_cache_scales(tf.name, a["tp_rank"], a["tp_size"], a["layers"], a["model_type"])
        finally:
            os.unlink(tf.name)


def test_rotary_tables_golden(golden_dir):
    """Host logic: the cos|sin table, plain and with Llama-3.1 (llama3) frequency scaling, vs the reference's
    RotaryEmbedding / Llama3RotaryEmbedding."""
    from aphrodite_engine_amd.model import _rope_cache
    g = np.load(os.path.join(golden_dir, "rope.npz"))
    plain = _rope_cache(128, 640, 500000.0, torch.float32, "cpu")
    np.testing.assert_array_equal(plain.numpy(), g["plain"])
    scaled = _rope_cache(128, 640, 500000.0, torch.float32, "cpu",
                         {"rope_type": "llama3", "factor": 8.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0,
                          "original_max_position_embeddings": 8192})
    np.testing.assert_array_equal(scaled.numpy(), g["llama3"])
    assert not np.array_equal(g["plain"], g["llama3"])
    with pytest.raises(NotImplementedError):
        _rope_cache(128, 16, 10000.0, torch.float32, "cpu", {"rope_type": "longrope", "factor": 4.0})


def test_scaled_rotary_tables_golden(golden_dir):
    """Host logic: the linear, dynamic-NTK and YaRN tables (rows, frequencies, mscale) vs the reference's
    LinearScaling / DynamicNTKScaling / YaRNScalingRotaryEmbedding (tests/golden/make_golden_rope_scaling.py), bit for bit."""
    import json
    from aphrodite_engine_amd.model import _rope_cache
    g = np.load(os.path.join(golden_dir, "rope_scaling.npz"))
    cases = json.load(open(os.path.join(golden_dir, "rope_scaling_cases.json")))
    assert set(cases) == set(g.files) == {"linear", "dynamic", "yarn", "yarn_kw"}
    for tag, c in cases.items():
        got = _rope_cache(c["head_dim"], c["max_pos"], c["theta"], torch.float32, "cpu", c["rope_scaling"])
        assert got.shape == g[tag].shape, tag
        np.testing.assert_array_equal(got.numpy(), g[tag], err_msg=tag)
    assert not np.array_equal(g["yarn"], g["yarn_kw"])


def test_compressed_tensors_scheme_selection_golden(golden_dir):
    """Host logic: CompressedTensorsConfig.get_quant_method vs the reference's _get_scheme_from_parts on its own
    pydantic QuantizationArgs.  Schemes outside the hot path (int8 W8A8, 2:4 sparse) must be
    refused, everything else (FP8 W8A8 / W8A16, 4- and 8-bit WNA16) must select the same scheme with the same arguments."""
    import json
    from aphrodite_engine_amd.quantization.compressed_tensors import (CompressedTensorsConfig,
                                                                      CompressedTensorsW8A16Fp8Method,
                                                                      CompressedTensorsWNA16Method)
    from aphrodite_engine_amd.quantization.fp8 import CompressedTensorsW8A8Fp8Method
    cases = json.load(open(os.path.join(golden_dir, "ct_schemes.json")))
    assert len(cases) >= 15
    seen, bits_seen = set(), set()
    for c in cases:
        cfg = CompressedTensorsConfig.from_config({"format": c["format"], "config_groups": {"group_0": {
            "targets": ["Linear"], "weights": c["weights"], "input_activations": c["input_activations"]}}})
        want = c["result"]
        outside = want == "NotImplementedError" or want["scheme"] in ("W4A16Sparse24", "W8A8Int8")
        if outside:
            with pytest.raises(NotImplementedError):
                cfg.get_quant_method(torch.nn.Module(), "model.layers.0.mlp.down_proj")
            continue
        m = cfg.get_quant_method(torch.nn.Module(), "model.layers.0.mlp.down_proj")
        seen.add(want["scheme"])
        if want["scheme"] == "W8A8Fp8":
            assert isinstance(m, CompressedTensorsW8A8Fp8Method)
            assert m.quant_config.strategy == want["strategy"]
            assert m.quant_config.is_static_input_scheme == bool(want["is_static_input_scheme"])
        elif want["scheme"] == "W8A16Fp8":
            assert isinstance(m, CompressedTensorsW8A16Fp8Method) and m.strategy == want["strategy"]
        else:
            assert isinstance(m, CompressedTensorsWNA16Method)
            assert m.pack_factor == 32 // want["num_bits"] and m.quant_type.size_bits == want["num_bits"]
            bits_seen.add(want["num_bits"])
            assert m.strategy == want["strategy"]
            assert m.group_size == (-1 if want["group_size"] is None else want["group_size"])
            assert m.has_g_idx == (want["actorder"] == "group")
    assert seen == {"W8A8Fp8", "W8A16Fp8", "WNA16"} and bits_seen == {4, 8}


def test_parameter_contracts_golden(golden_dir):
    """Host logic: the parameters every quant method registers (names, shapes, dtypes and the loader metadata
    input_dim / output_dim / packed_dim / pack factor / one-scalar-per-shard) vs what the reference's own
    create_weights registers, over column/row-parallel geometries at TP 1 and 2 (44 cases)."""
    import json
    from aphrodite_engine_amd.quantization.awq import AWQConfig
    from aphrodite_engine_amd.quantization.fp8 import (CompressedTensorsW8A8Fp8Config, CompressedTensorsW8A8Fp8Method,
                                                       Fp8Config)
    from aphrodite_engine_amd.quantization.gptq import GPTQConfig
    cases = json.load(open(os.path.join(golden_dir, "param_contracts.json")))
    assert len(cases) == 44
    for c in cases:
        kin, outs, kfull, nfull = c["args"]
        cfg = c["config"]
        layer = torch.nn.Module()
        if c["method"] == "gptq":
            method = GPTQConfig(4, cfg["group_size"], cfg["desc_act"]).get_quant_method(layer, "")
            dtype = torch.float16
        elif c["method"] == "awq":
            method = AWQConfig(4, cfg["group_size"], True).get_quant_method(layer, "")
            dtype = torch.float16
        elif c["method"] == "fp8":
            method = Fp8Config(cfg["serialized"], cfg["activation_scheme"]).get_quant_method(layer, "")
            dtype = torch.bfloat16
        else:
            method = CompressedTensorsW8A8Fp8Method(CompressedTensorsW8A8Fp8Config(cfg["strategy"], cfg["static"]))
            dtype = torch.bfloat16
        method.create_weights(layer, kin, outs, kfull, nfull, dtype, weight_loader=None)
        mine = dict(layer.named_parameters())
        for name, want in c["params"].items():
            if want is None:
                assert getattr(layer, name, None) is None, (c["method"], c["geom"], name)
                continue
            prm = mine.pop(name)
            tag = (c["method"], c["geom"], cfg, name)
            assert list(prm.shape) == want["shape"], tag
            assert str(prm.dtype) == want["dtype"], tag
            for key in ("input_dim", "output_dim", "packed_dim"):
                assert getattr(prm, key, None) == want.get(key), tag + (key, )
            assert getattr(prm, "pack_factor", None) == want.get("packed_factor"), tag
            assert bool(getattr(prm, "needs_scalar_to_array", False)) == (want["kind"] == "PerTensorScaleParameter"), tag
        assert not mine, f"extra parameters {list(mine)} in {c['method']} {c['geom']}"
        if "exllama_state" in c:
            assert layer.exllama_state.name == c["exllama_state"]


# ---------------------------------------------------------------------------
# glue restatements vs the reference's own compiled CPU kernels (oracle/_ref)
# ---------------------------------------------------------------------------
def test_glue_ops_vs_reference_cpu_kernels():
    """oracle rms_norm / fused_add_rms_norm / silu_and_mul / rotary_embedding (fp64 math) vs kernels/cpu/
    {layernorm,activation,pos_encoding}.cpp compiled where they lie, on fp32 inputs."""
    ops = _ref_lib()
    torch.manual_seed(3)
    T, H = 7, 256
    x = torch.randn(T, H)
    w = torch.rand(H) + 0.5
    out = torch.empty_like(x)
    ops.rms_norm(out, x, w, 1e-5)
    np.testing.assert_allclose(out.numpy(), oa.rms_norm(x.numpy(), w.numpy(), 1e-5), rtol=2e-6, atol=2e-6)
    xi, res = torch.randn(T, H), torch.randn(T, H)
    x2, r2 = xi.clone(), res.clone()
    ops.fused_add_rms_norm(x2, r2, w, 1e-5)
    o_ref, r_ref = oa.fused_add_rms_norm(xi.numpy(), res.numpy(), w.numpy(), 1e-5)
    np.testing.assert_allclose(r2.numpy(), r_ref, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(x2.numpy(), o_ref, rtol=2e-6, atol=2e-6)
    gu = torch.randn(T, 2 * 96) * 2
    act = torch.empty(T, 96)
    ops.silu_and_mul(act, gu)
    np.testing.assert_allclose(act.numpy(), oa.silu_and_mul(gu.numpy()), rtol=2e-6, atol=2e-6)
    # NeoX rotary on q [T, 4 heads x 64] and k [T, 2 heads x 64]
    hd, max_pos = 64, 97
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2).float() / hd))
    fr = torch.einsum("i,j->ij", torch.arange(max_pos).float(), inv)
    cs = torch.cat((fr.cos(), fr.sin()), dim=-1)
    pos = torch.tensor([0, 5, 96, 17, 1, 33, 64])
    q, k = torch.randn(T, 4 * hd), torch.randn(T, 2 * hd)
    q2, k2 = q.clone(), k.clone()
    ops.rotary_embedding(pos, q2, k2, hd, cs, True)
    qr, kr = oa.rotary_embedding_neox(pos.numpy(), q.numpy(), k.numpy(), hd, cs.numpy())
    np.testing.assert_allclose(q2.numpy(), qr, rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(k2.numpy(), kr, rtol=2e-6, atol=2e-6)


def test_copy_blocks_vs_reference_cpu_kernel():
    ops = _ref_lib()
    torch.manual_seed(4)
    layers, NB = 3, 10
    kcs = [torch.randn(NB, 2, 8, 16, 4) for _ in range(layers)]
    vcs = [torch.randn(NB, 2, 32, 16) for _ in range(layers)]
    mapping = torch.tensor([[0, 5], [0, 7], [3, 9], [8, 1]], dtype=torch.int64)
    k2 = [k.numpy().copy() for k in kcs]
    v2 = [v.numpy().copy() for v in vcs]
    ops.copy_blocks(kcs, vcs, mapping)
    oa.copy_blocks(k2, v2, mapping.numpy())
    for a, b in zip(kcs + vcs, k2 + v2):
        np.testing.assert_array_equal(a.numpy(), b)


# ---- GPTQ exllama path: pinned by the REFERENCE's own CUDA kernels run on the host -----------------------------
# tests/golden/gptq_ref.npz holds what kernels/quantization/gptq/q_gemm.cu's shuffle_4bit_kernel /
# make_sequential_4bit_kernel / reconstruct_exllama_4bit_kernel / reconstruct_gptq_kernel /
# gemm_half_q_half_gptq_4bit_kernel return when compiled for the CPU (oracle/Makefile, oracle/ref_gptq_bind.cpp);
# generated by tests/golden/make_golden_gptq.py.  Integer work and the fp16 dequant are bit-exact; the GEMM is
# compared within the reference kernel's own fp16-dot / fp16-atomics noise.
@pytest.fixture(scope="module")
def gref(golden_dir):
    return np.load(os.path.join(golden_dir, "gptq_ref.npz"))


def _gref_case(gref, name):
    perm = gref[f"{name}_perm"]
    return (gref[f"{name}_qweight"], gref[f"{name}_qzeros"], gref[f"{name}_scales"], gref[f"{name}_g_idx"],
            perm if len(perm) else None)


@pytest.mark.parametrize("name", ["a", "b", "c", "d"])
def test_gptq_shuffle_reference_kernels(gref, name):
    qw, qz, sc, g_idx, perm = _gref_case(gref, name)
    got = oq.gptq_shuffle(qw, perm)
    np.testing.assert_array_equal(got.view(np.uint32), gref[f"{name}_shuffle"].view(np.uint32))


@pytest.mark.parametrize("name", ["a", "b", "c", "d"])
def test_gptq_dequant_reference_kernels(gref, name):
    qw, qz, sc, g_idx, perm = _gref_case(gref, name)
    # (q - (z + 1)) * s from checkpoint order + g_idx: reconstruct_gptq_kernel, q_gemm.cu:1394-1434
    w = oq.gptq_dequant(qw, qz, sc, g_idx, shuffled=False).astype(np.float16)
    np.testing.assert_array_equal(w.view(np.uint16), gref[f"{name}_recon_gptq"].view(np.uint16))
    # the same weights through the exllama layout: reconstruct_exllama_4bit_kernel writes row perm[k] (:937-960)
    shuf = gref[f"{name}_shuffle"]
    w_seq = oq.gptq_dequant(shuf, qz, sc, None, shuffled=True).astype(np.float16)
    full = np.zeros_like(w_seq)
    if perm is not None:
        full[perm, :] = w_seq
    else:
        full = w_seq
    np.testing.assert_array_equal(full.view(np.uint16), gref[f"{name}_recon_exl"].view(np.uint16))
    np.testing.assert_array_equal(full.view(np.uint16), w.view(np.uint16))   # both reference kernels agree, too


@pytest.mark.parametrize("name", ["c", "d"])
@pytest.mark.parametrize("m", [1, 5, 8, 13])
def test_gptq_gemm_reference_kernel(gref, name, m):
    qw, qz, sc, g_idx, perm = _gref_case(gref, name)
    a = gref[f"{name}_gemm_a{m}"]
    ref = gref[f"{name}_gemm_c{m}"].astype(np.float64)
    got = oq.gptq_gemm(a, gref[f"{name}_shuffle"], qz, sc, perm, True)
    # the reference kernel: fp16 hfma2 dot over 8 k, fp32 per 128-k block, fp16 atomics across blocks
    assert np.isfinite(ref).all()
    assert np.abs(ref - got).max() <= 2e-2
    assert np.abs(ref - got).mean() / np.abs(got).mean() < 2e-3


def test_gptq_reference_library_live(gref):
    """When oracle/_ref/libaphro_ref_gptq.so is present (build container, GPU box), run the reference kernels NOW on
    fresh random inputs -- the fixtures above were not cherry-picked."""
    import ctypes
    import importlib.util
    lib_path = os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref", "libaphro_ref_gptq.so")
    if not os.path.exists(lib_path):
        pytest.skip("oracle/_ref/libaphro_ref_gptq.so not built")
    spec = importlib.util.spec_from_file_location("make_golden_gptq", os.path.join(os.path.dirname(__file__), "golden",
                                                                                   "make_golden_gptq.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    lib = ctypes.CDLL(lib_path)
    rng = np.random.default_rng(7)
    for k, n, gs, ao in [(128, 64, 32, True), (384, 128, 128, False), (256, 32, 64, True)]:
        qw, qz, sc, g_idx, perm = mg.make_case(rng, k, n, gs, ao)
        shuf = mg.ref_shuffle(lib, qw, perm)
        np.testing.assert_array_equal(oq.gptq_shuffle(qw, perm).view(np.uint32), shuf.view(np.uint32))
        w = oq.gptq_dequant(qw, qz, sc, g_idx, shuffled=False).astype(np.float16)
        np.testing.assert_array_equal(w.view(np.uint16), mg.ref_recon_gptq(lib, qw, qz, sc, g_idx).view(np.uint16))
        a = rng.standard_normal((3, k)).astype(np.float16)
        c = mg.ref_gemm(lib, a, shuf, qz, sc, perm).astype(np.float64)
        assert np.abs(c - oq.gptq_gemm(a, shuf, qz, sc, perm, True)).max() <= 2e-2


# ---- GPTQ 2 / 3 / 8-bit: the checkpoint layout pinned by the reference's reconstruct_gptq kernels run on the host ------
@pytest.mark.parametrize("bits", [2, 3, 8])
@pytest.mark.parametrize("name", ["p", "q"])
def test_gptq_dequant_bits_reference_kernels(golden_dir, bits, name):
    """reconstruct_gptq_kernel<MatrixView_q{2,8}_row> / reconstruct_gptq_3bit_kernel (q_gemm.cu:1394-1505): bit-exact,
    including the 3-bit values / zero points that straddle a word and the unmasked `zero + 1` (8-bit 255 -> 256)."""
    g = np.load(os.path.join(golden_dir, "gptq_ref_bits.npz"))
    k = f"b{bits}{name}"
    w = oq.gptq_dequant(g[k + "_qweight"], g[k + "_qzeros"], g[k + "_scales"], g[k + "_g_idx"], shuffled=False,
                        bits=bits).astype(np.float16)
    np.testing.assert_array_equal(w.view(np.uint16), g[k + "_recon_gptq"].view(np.uint16))


@pytest.mark.parametrize("bits", [2, 3, 8])
def test_gptq_bits_pack_roundtrip_and_shuffle(bits):
    rng = np.random.default_rng(bits)
    k, n, gs = 128, 64, 32
    q = rng.integers(0, 1 << bits, size=(k, n))
    z = rng.integers(0, 1 << bits, size=(k // gs, n))
    qw, qz = oq.gptq_pack(q, bits), oq.pack_cols(z, bits)
    assert qw.shape == (k * bits // 32, n) and qz.shape == (k // gs, n * bits // 32)
    np.testing.assert_array_equal(oq.gptq_unpack(qw, bits), q)
    np.testing.assert_array_equal(oq.unpack_cols(qz, bits), z)
    # act-order: W through (shuffle with perm, gather a[:, perm]) == W through g_idx on the checkpoint order
    g_idx = rng.permutation(np.arange(k) // gs).astype(np.int32)
    perm = np.argsort(g_idx, kind="stable").astype(np.int32)
    sc = rng.uniform(0.002, 0.02, size=(k // gs, n)).astype(np.float16)
    a = rng.standard_normal((3, k)).astype(np.float16)
    y0 = oq.gptq_gemm(a, qw, qz, sc, g_idx, False, bits)
    y1 = oq.gptq_gemm(a, oq.gptq_shuffle(qw, perm, bits), qz, sc, perm, True, bits)
    np.testing.assert_allclose(y0, y1, rtol=1e-12, atol=1e-12)
