"""Round 6: the dynamic per-token FP8 scheme without its two quantising launches.

The reference quantises activations in a launch of its own (``dynamic_per_token_scaled_fp8_quant``,
kernels/quantization/fp8/common.cu:201-256: scale = max(absmax / 448, 1 / (448 * 512)), q = fp8(x / scale) with a true
division "to match FBGemm", :44-52) in front of every W8A8 GEMM (quantization/utils/w8a8_utils.py:104-126).  Here the
producers (attention, gate_up + SiluAndMul epilogue) leave 16-bit activations plus absmax PARTIALS, and the consuming GEMM
reduces the partials and quantises its A fragments on load with a hoisted-reciprocal form of the IEEE division.  Bit-exact
or it is not done:

* the quantiser against the CPU oracle (oracle/fp8.py) for ALL 65 536 f16 / bf16 inputs x every kind of scale, and against
  x / scale for EVERY (input, absmax) pair that can occur (absmax is itself one of the row's values: 32 768 x 65 536 pairs);
* the fused GEMM forms against the op sequences they stand for, bit for bit;
* the decoder layer with and without the diet, bit for bit (and, in test_headline_gpu.py, against the oracle-composed layer).
"""
import dataclasses
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from aphrodite_engine_amd import _custom_ops
    return _custom_ops


def _all_values(dtype):
    """every finite 16-bit pattern of dtype, as a [65536]-tensor (NaN / Inf patterns replaced by 0)."""
    bits = torch.arange(65536, dtype=torch.int32).to(torch.int16)
    x = bits.view(dtype)
    return torch.where(torch.isfinite(x.float()), x, torch.zeros_like(x))


def _special_absmax(dtype):
    """absmax values (magnitudes of 16-bit values) that make every kind of scale: the floor 1 / (448 * 512), quotients that are
    fp8-subnormal, powers of two and their neighbours, the largest finite value, and a random spread."""
    mags = _all_values(dtype).float().abs().unique()
    mags = mags[mags > 0]
    floor_abs = 448.0 / (448.0 * 512.0)
    pick = [mags[0], mags[1], mags[mags <= floor_abs][-1], mags[mags > floor_abs][0], mags[-1], mags[-2]]
    for e in range(-14, 16):
        near = (mags - 2.0 ** e).abs().argmin().item()
        pick += [mags[max(near - 1, 0)], mags[near], mags[min(near + 1, len(mags) - 1)]]
    g = torch.Generator().manual_seed(6)
    pick += list(mags[torch.randint(0, len(mags), (900, ), generator=g)])
    return torch.stack([torch.as_tensor(float(p)) for p in pick]).unique()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_quantise_on_load_all_inputs_vs_cpu_oracle(ops, dtype):
    """All 65 536 inputs x ~1000 absmax values (floor, subnormal quotients, power-of-two edges, largest, random) against the
    CPU oracle's dynamic_per_token_scaled_fp8_quant, on the domain the kernel sees: |x| <= the row's absmax.  Bits and scales."""
    from oracle import fp8 as of8
    vals = _all_values(dtype)
    amax = _special_absmax(dtype)
    M = len(amax)
    # a row = every value no larger than its absmax (the rest zeroed: they cannot occur in a row with that absmax)
    x = vals[None, :].repeat(M, 1)
    x = torch.where(x.float().abs() <= amax[:, None], x, torch.zeros_like(x))
    # the partials: the absmax somewhere among smaller partials, np = 8 (attention) -- the reduce must find it
    part = torch.zeros(M, 8)
    part[torch.arange(M), torch.arange(M) % 8] = amax
    part[torch.arange(M), (torch.arange(M) + 3) % 8] = amax * 0.5
    q, sc = ops.fp8_quant_rows_aq(x.to(DEV).contiguous(), part.to(DEV).contiguous())
    want_q, want_s = of8.dynamic_per_token_scaled_fp8_quant(x.float().numpy())
    np.testing.assert_array_equal(sc.cpu().numpy(), want_s)
    got = q.view(torch.uint8).cpu().numpy()
    bad = np.argwhere(got != want_q)
    assert bad.size == 0, (len(bad), bad[:5], [(float(x[i, j]), float(amax[i]), int(got[i, j]), int(want_q[i, j])) for i, j in bad[:5]])
    # and the op it replaces, on the same rows
    q_op, s_op = ops.scaled_fp8_quant(x.to(DEV).contiguous(), None, use_per_token_if_dynamic=True)
    assert torch.equal(q_op.view(torch.uint8).cpu(), q.view(torch.uint8).cpu()) and torch.equal(s_op.cpu(), sc.cpu())


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_quantise_on_load_every_input_absmax_pair(ops, dtype):
    """EVERY pair that can occur: absmax is one of the row's own values, so the scales are the <= 32 768 values
    fl(a / 448) (floored) over the 16-bit magnitudes a, and a row may hold any x with |x| <= a: 32 768 x 65 536 pairs, in
    chunks, against fp8(clamp(x / scale)) computed by torch on the device -- an expression the first assertion pins to the CPU
    oracle on this very device before it is trusted."""
    from oracle import fp8 as of8
    vals = _all_values(dtype).to(DEV)
    mags = vals.float().abs().unique()

    def ref(xr, s):
        return (xr.float() / s).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8)

    def scales_of(a):      # fl(a / 448) floored, in numpy on the host: torch folds a division by a Python scalar into a reciprocal multiply
        return torch.from_numpy(np.maximum(a.cpu().numpy().astype(np.float32) / np.float32(448.0),
                                           np.float32(1.0) / (np.float32(448.0) * np.float32(512.0)))).to(DEV)[:, None]

    probe_a = mags[torch.randint(0, len(mags), (64, ), device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))]
    xr = torch.where(vals.float().abs()[None, :] <= probe_a[:, None], vals[None, :], torch.zeros_like(vals)[None, :])
    s = scales_of(probe_a)
    want_q, want_s = of8.dynamic_per_token_scaled_fp8_quant(xr.float().cpu().numpy())
    np.testing.assert_array_equal(s.cpu().numpy(), want_s)
    np.testing.assert_array_equal(ref(xr, s).cpu().numpy(), want_q)          # the device expression IS the oracle's
    CH = 512
    for lo in range(0, len(mags), CH):
        a = mags[lo:lo + CH]
        xr = torch.where(vals.float().abs()[None, :] <= a[:, None], vals[None, :], torch.zeros_like(vals)[None, :]).contiguous()
        part = torch.zeros(len(a), 4, device=DEV)
        part[:, lo // CH % 4] = a
        q, sc = ops.fp8_quant_rows_aq(xr, part)
        s = scales_of(a)
        assert torch.equal(sc, s), lo
        w = ref(xr, s)
        if not torch.equal(q.view(torch.uint8), w):
            bad = (q.view(torch.uint8) != w).nonzero()[:5]
            raise AssertionError([(float(xr[i, j]), float(a[i]), int(q.view(torch.uint8)[i, j]), int(w[i, j])) for i, j in bad])


def _fp8_weight(n, k, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return (torch.randn(n, k, device=DEV, generator=g) * 0.5).to(torch.float8_e4m3fn)


def _adversarial_rows(m, k, dtype, seed):
    """activations with everything the quantiser must get right inside real rows: -0, 16-bit subnormals, an all-zero row
    (floor scale), a row whose absmax is tiny, a huge outlier, values exactly at +-absmax."""
    g = torch.Generator(device=DEV).manual_seed(seed)
    x = torch.randn(m, k, device=DEV, generator=g) * torch.logspace(-3, 2, m, device=DEV)[:, None]
    x = x.to(dtype)
    x[:, 1::97] = -0.0
    x[:, 5::131] = torch.finfo(dtype).smallest_normal / 4 if dtype == torch.float16 else 1e-39
    if m > 1:
        x[1] = 0
    if m > 2:
        x[2] = (x[2].float() * 1e-6).to(dtype)
    if m > 3:
        x[3, 7] = 60000.0 if dtype == torch.float16 else 3e38
    x[0, 11] = -x[0].float().abs().max().to(dtype)
    return x.contiguous()


def _to_pairs(ops, x):
    """[M, K] -> the flat pair-major buffer (rows of the last 16-row tile beyond M poisoned with NaN: never read back)."""
    m, k = x.shape
    buf = torch.full((ops.aq_pairs_numel(m, k), ), float("nan"), dtype=x.dtype, device=x.device)
    buf[ops.aq_pairs_index(m, k, x.device).flatten()] = x.flatten()
    return buf


@pytest.mark.parametrize("pairs", [False, True])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M", [1, 7, 16, 17, 32])
@pytest.mark.parametrize("N,K,np_", [(4096, 4096, 8), (4096, 14336, 256), (6144, 4096, 4), (4096, 14336, 64)])
def test_gemm_quantise_on_load_matches_quant_then_gemm(ops, N, K, np_, M, dtype, pairs):
    """ops.fp8_gemm_resident_aq(x16, partials) == ops.scaled_fp8_quant(x16, per token) -> ops.fp8_gemm_resident, bit for bit:
    fp32 slabs (every K slice) and scales, at the o_proj / down / qkv plans of Llama-3-8B, ragged M, adversarial rows; A
    row-major or in the pair-major layout the fused producers write (rows beyond M poisoned)."""
    if ops.fp8_gemm_resident_ksplit(M, N, K) <= 0:
        pytest.skip("shape not served")
    strip = ops.fp8_strip_relayout(_fp8_weight(N, K, 3), M)
    x = _adversarial_rows(M, K, dtype, 11 + M)
    part = x.float().abs().view(M, np_, K // np_).amax(dim=2).contiguous()
    q, s = ops.scaled_fp8_quant(x, None, use_per_token_if_dynamic=True)
    want = ops.fp8_gemm_resident(q, strip, slabs=True)
    got, sc = ops.fp8_gemm_resident_aq(_to_pairs(ops, x) if pairs else x, part, strip, a_pairs=pairs)
    assert torch.equal(sc, s)
    assert torch.equal(got.view(torch.int32), want.view(torch.int32))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M", [1, 16, 23, 32])
def test_gate_up_silu_epilogue_and_down_match_the_op_sequence(ops, M, dtype):
    """gate_up + SiluAndMul epilogue on the interleaved strip copy == fp8_gemm_resident (scaled epilogue) -> silu_and_mul:
    the 16-bit activation bit for bit, the partials' maximum == the row absmax; then the down projection on the
    quantise-on-load form == silu_and_mul_quant_fp8 -> fp8_gemm_resident (slabs + scales).  Static scheme: the e4m3 output."""
    N, K, I = 28672, 4096, 14336
    w = _fp8_weight(N, K, 5)
    g = torch.Generator(device=DEV).manual_seed(17)
    x = (torch.randn(M, K, device=DEV, generator=g) * 2).to(dtype)
    qx, sx = ops.scaled_fp8_quant(x, None, use_per_token_if_dynamic=True)
    ws = (torch.rand(N, 1, device=DEV, generator=g) * 0.02 + 0.002)
    gate_up = ops.fp8_gemm_resident(qx, ops.fp8_strip_relayout(w, M), sx, ws, out_dtype=dtype)
    q_ref, s_ref, act_ref = ops.silu_and_mul_quant_fp8(gate_up, want_out=True)
    il = ops.fp8_strip_relayout_interleaved(w, M)
    act, part = ops.fp8_gemm_resident_silu(qx, il, sx, ws, dtype)
    assert part.shape == (M, ops.fp8_gemm_resident_strips(M, N, K))
    assert torch.equal(act.view(torch.int16), act_ref.view(torch.int16))
    assert torch.equal(part.amax(dim=1), act_ref.float().abs().amax(dim=1))
    wd = ops.fp8_strip_relayout(_fp8_weight(4096, I, 9), M)
    got, sd = ops.fp8_gemm_resident_aq(act, part, wd)
    assert torch.equal(sd, s_ref)
    want_slabs = ops.fp8_gemm_resident(q_ref, wd, slabs=True)
    assert torch.equal(got.view(torch.int32), want_slabs.view(torch.int32))
    # the step's form: the activation leaves the epilogue pair-major and the down GEMM reads it lane-linearly
    act_p, part_p = ops.fp8_gemm_resident_silu(qx, il, sx, ws, dtype, act_pairs=True)
    assert torch.equal(part_p, part)
    assert torch.equal(act_p[ops.aq_pairs_index(M, I, DEV)].view(torch.int16), act_ref.view(torch.int16))
    got_p, sd_p = ops.fp8_gemm_resident_aq(act_p, part_p, wd, a_pairs=True)
    assert torch.equal(sd_p, s_ref) and torch.equal(got_p.view(torch.int32), want_slabs.view(torch.int32))
    st = torch.tensor([0.037], device=DEV)
    q_st = ops.fp8_gemm_resident_silu(qx, il, sx, ws, dtype, static_out_scale=st)
    assert torch.equal(q_st.view(torch.uint8), ops.silu_and_mul_quant_fp8(gate_up, static_scale=st)[0].view(torch.uint8))


def test_gate_up_silu_epilogue_vs_oracle(ops):
    """... and against the ORACLE's composition (scaled_mm -> rounding to bf16 -> silu_and_mul), not only against this
    package's own op sequence."""
    from oracle import attention as oa
    from oracle import fp8 as of8
    M, N, K = 8, 28672, 4096
    w = _fp8_weight(N, K, 21)
    g = torch.Generator(device=DEV).manual_seed(2)
    x = torch.randn(M, K, device=DEV, generator=g).to(torch.bfloat16)
    ws = (torch.rand(N, 1, device=DEV, generator=g) * 0.02 + 0.002)
    qx, sx = of8.dynamic_per_token_scaled_fp8_quant(x.float().cpu().numpy())
    gu = of8.scaled_mm(qx, w.t().view(torch.uint8).cpu().numpy(), sx, ws.cpu().numpy())
    gu = torch.from_numpy(gu).to(torch.bfloat16).float().numpy()
    want = torch.from_numpy(oa.silu_and_mul(gu)).to(torch.bfloat16).float().numpy()
    q_dev, s_dev = ops.scaled_fp8_quant(x, None, use_per_token_if_dynamic=True)
    act, _ = ops.fp8_gemm_resident_silu(q_dev, ops.fp8_strip_relayout_interleaved(w, M), s_dev, ws, torch.bfloat16)
    got = act.float().cpu().numpy()
    assert np.abs(got - want).mean() <= 2e-3 * np.abs(want).mean() + 1e-6
    np.testing.assert_allclose(got, want, rtol=2e-2, atol=2e-2 * np.abs(want).max())


@pytest.mark.parametrize("kv_cache_dtype", ["auto", "fp8"])
def test_attention_absmax_partials(ops, kv_cache_dtype):
    """The attention launch's 16-bit output is unchanged by asking for the partials, and partial[seq][kv-head] is the absmax of
    that (sequence, kv-head) slice of it."""
    from aphrodite_engine_amd import model as M
    from aphrodite_engine_amd.attention.paged_attn import PagedAttention
    cfg = dataclasses.replace(M.LLAMA3_8B, num_hidden_layers=1, vocab_size=512, max_position_embeddings=2048)
    lens = [1, 16, 17, 700, 1024, 333, 5]
    bs = len(lens)
    dtype = torch.bfloat16
    meta, pos, nblocks = M.make_decode_metadata(bs, lens, 16, DEV)
    hq, hkv, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    g = torch.Generator(device=DEV).manual_seed(4)
    slabs = torch.randn(2, bs, (hq + 2 * hkv) * hd, device=DEV, generator=g) * 0.3
    row = torch.rand(bs, 1, device=DEV, generator=g) + 0.5
    col = torch.rand((hq + 2 * hkv) * hd, device=DEV, generator=g) + 0.5
    cos_sin = M._rope_cache(hd, 2048, cfg.rope_theta, dtype, torch.device(DEV), None)
    outs = []
    for want_absmax in (False, True):
        kc, vc = PagedAttention.split_kv_cache(M.make_kv_caches(cfg, nblocks, 16, dtype, kv_cache_dtype, DEV, seed=5)[0], hkv, hd)
        outs.append(ops.paged_attention_rope_scaled(slabs, row, col, pos, cos_sin, meta.slot_mapping, kc, vc, hq, hkv, hd ** -0.5,
                                                    meta.block_tables, meta.seq_lens_tensor, 16, max(lens), None, kv_cache_dtype,
                                                    0.5, 0.25, want_absmax=want_absmax))
        if want_absmax:
            kc, vc = PagedAttention.split_kv_cache(M.make_kv_caches(cfg, nblocks, 16, dtype, kv_cache_dtype, DEV, seed=5)[0], hkv, hd)
            outs.append(ops.paged_attention_rope_scaled(slabs, row, col, pos, cos_sin, meta.slot_mapping, kc, vc, hq, hkv, hd ** -0.5,
                                                        meta.block_tables, meta.seq_lens_tensor, 16, max(lens), None, kv_cache_dtype,
                                                        0.5, 0.25, want_absmax=True, out_pairs=True))
    out, (out2, part), (out_p, part_p) = outs
    assert torch.equal(out.view(torch.int16), out2.view(torch.int16))
    assert torch.equal(part, out2.float().abs().view(bs, hkv, -1).amax(dim=2))
    assert torch.equal(part_p, part)
    assert torch.equal(out_p[ops.aq_pairs_index(bs, hq * hd, DEV)].view(torch.int16), out.view(bs, hq * hd).view(torch.int16))


@pytest.mark.parametrize("kv_cache_dtype", ["auto", "fp8"])
@pytest.mark.parametrize("bs", [4, 32])
def test_fused_fp8_decode_layers_with_and_without_the_diet_bit_for_bit(ops, bs, kv_cache_dtype):
    """Two Llama-3-8B-geometry layers of the dynamic per-token scheme: 7 launches per layer (round 6) against the 9-launch form
    (APHRO_FP8_NO_LAUNCH_DIET=1) -- hidden states and KV caches identical to the bit."""
    from aphrodite_engine_amd import model as M
    from aphrodite_engine_amd.quantization.fp8 import CompressedTensorsW8A8Fp8Config
    cfg = dataclasses.replace(M.LLAMA3_8B, num_hidden_layers=2, vocab_size=2048, max_position_embeddings=2048)
    rng = np.random.default_rng(bs)
    lens = [int(v) for v in rng.integers(1, 1500, size=bs)]
    dtype = torch.bfloat16
    with torch.no_grad():
        m = M.LlamaForCausalLM(cfg, CompressedTensorsW8A8Fp8Config("channel", is_static_input_scheme=False), dtype, kv_cache_dtype)
        m.init_synthetic(torch.device(DEV))
        m.use_fused_decode = True
        meta, pos, nblocks = M.make_decode_metadata(bs, lens, 16, DEV)
        ids = torch.randint(0, cfg.vocab_size, (bs, ), device=DEV, generator=torch.Generator(device=DEV).manual_seed(7))
        res = []
        for diet in (True, False):
            if not diet:
                os.environ["APHRO_FP8_NO_LAUNCH_DIET"] = "1"
            try:
                for layer in m.layers:
                    layer.enable_fp8_strips(bs)
                assert (m.layers[0].fp8_gate_up_il is not None) == diet
                caches = M.make_kv_caches(cfg, nblocks, 16, dtype, kv_cache_dtype, DEV, seed=5)
                hidden = m(ids, pos, caches, meta)
                res.append((hidden.clone(), [c.clone() for c in caches]))
            finally:
                os.environ.pop("APHRO_FP8_NO_LAUNCH_DIET", None)
        (h1, c1), (h0, c0) = res
        assert torch.isfinite(h1.float()).all()
        assert torch.equal(h1.view(torch.int16), h0.view(torch.int16))
        for a, b in zip(c1, c0):
            assert torch.equal(a.view(torch.uint8), b.view(torch.uint8))
