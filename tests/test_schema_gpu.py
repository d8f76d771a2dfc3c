"""The drop-in boundary AS THE REFERENCE SEES IT: torch.library ops with the reference's schemas
(kernels/torch_bindings.cpp, kernels/rocm/torch_bindings.cpp -- SURVEY.md 8b "the contract is the schema"), dispatched
with device tensors, eagerly and under HIP-graph capture + replay (decode runs captured: worker/model_runner.py:1360-1507),
against the oracle.  Registered under private namespaces so that a real ``aphrodite._C`` could be loaded beside them."""
import numpy as np
import pytest
import torch

from oracle import attention as oa
from oracle import fp8 as of8
from oracle import quant as oq

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def T():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X (torch.cuda.is_available() is False)")
    from aphrodite_engine_amd import _lib, torch_ops
    _lib.lib()
    torch_ops._REGISTERED = False
    torch_ops.register("_aphro_g_C", "_aphro_g_cache", "_aphro_g_rocm", "_aphro_g_moe")

    class NS:
        C = torch.ops._aphro_g_C
        cache = torch.ops._aphro_g_cache
        rocm = torch.ops._aphro_g_rocm
        moe = torch.ops._aphro_g_moe
    return NS


def t(x, dtype=None):
    out = torch.from_numpy(np.ascontiguousarray(x)).to(DEV)
    return out.to(dtype) if dtype is not None else out


def captured(fn):
    """Run fn() once eagerly (warm-up on a side stream, as torch requires), capture it into a HIP graph, replay it
    and return (graph, result of the captured call)."""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    g.replay()
    torch.cuda.synchronize()
    return g, out


def make_gptq(rng, K, N, G):
    w = (rng.standard_normal((K, N)) * 0.02).astype(np.float16)
    _, q, s, zp = oq.quantize_weights(w, 4, G, zero_points=True)
    return oq.gptq_pack(q), oq.gptq_pack_zeros(zp), s.astype(np.float16)


def test_schema_gptq_gemm_and_shuffle(T):
    rng = np.random.default_rng(0)
    K, N, G, M = 1024, 256, 128, 32
    qweight, qzeros, scales = make_gptq(rng, K, N, G)
    qw = t(qweight)
    T.C.gptq_shuffle(qw, torch.empty(0, dtype=torch.int32, device=DEV), 4)            # Tensor! q_weight, in place
    np.testing.assert_array_equal(qw.cpu().numpy(), oq.gptq_shuffle(qweight))
    a = t(rng.standard_normal((M, K)).astype(np.float16))
    empty = torch.empty(0, dtype=torch.int32, device=DEV)
    args = (a, qw, t(qzeros), t(scales), empty, True, 4)
    ref = oq.gptq_gemm(a.cpu().numpy(), qw.cpu().numpy(), qzeros, scales, None, True)
    eager = T.C.gptq_gemm(*args)
    np.testing.assert_allclose(eager.float().cpu().numpy(), ref, rtol=2e-3, atol=2e-3 * np.abs(ref).max())
    g, out = captured(lambda: T.C.gptq_gemm(*args))
    assert torch.equal(out, eager)
    # replay reads the CURRENT contents of the captured input buffer
    a2 = rng.standard_normal((M, K)).astype(np.float16)
    a.copy_(t(a2))
    g.replay()
    torch.cuda.synchronize()
    ref2 = oq.gptq_gemm(a2, qw.cpu().numpy(), qzeros, scales, None, True)
    np.testing.assert_allclose(out.float().cpu().numpy(), ref2, rtol=2e-3, atol=2e-3 * np.abs(ref2).max())


def test_schema_awq_gemm_and_dequantize(T):
    rng = np.random.default_rng(1)
    K, N, G, M = 512, 128, 128, 16
    w = (rng.standard_normal((K, N)) * 0.02).astype(np.float16)
    _, q, s, zp = oq.quantize_weights(w, 4, G, zero_points=True)
    qw, qz = oq.awq_pack(q), oq.awq_pack(zp)
    a = rng.standard_normal((M, K)).astype(np.float16)
    # positional order of the C++ schema: (_in_feats, _kernel, _scaling_factors, _zeros, split_k_iters)
    args = (t(a), t(qw), t(s), t(qz), 8)
    ref = oq.awq_gemm(a, qw, s, qz)
    _, out = captured(lambda: T.C.awq_gemm(*args))
    np.testing.assert_allclose(out.float().cpu().numpy(), ref, rtol=2e-3, atol=2e-3 * np.abs(ref).max())
    deq = T.C.awq_dequantize(t(qw), t(s), t(qz), 0, 0, 0)
    np.testing.assert_array_equal(deq.cpu().numpy(), oq.awq_dequantize(qw, s, qz))


@pytest.mark.parametrize("kv_cache_dtype", ["auto", "fp8"])
def test_schema_paged_attention_and_cache_write(T, kv_cache_dtype):
    rng = np.random.default_rng(2)
    S, Hq, Hkv, D, BS = 4, 8, 2, 128, 16
    dtype = torch.float16
    seq_lens = np.array([1, 40, 513, 700], np.int32)
    max_len = int(seq_lens.max())
    bps = (max_len + BS - 1) // BS
    NB = S * bps + 2
    x = 8 if kv_cache_dtype == "auto" else 16
    cdt = dtype if kv_cache_dtype == "auto" else torch.uint8
    kc = torch.zeros((NB, Hkv, D // x, BS, x), dtype=cdt, device=DEV)
    vc = torch.zeros((NB, Hkv, D, BS), dtype=cdt, device=DEV)
    kc_ref, vc_ref = kc.cpu().numpy().copy(), vc.cpu().numpy().copy()
    bt = rng.permutation(NB)[:S * bps].reshape(S, bps).astype(np.int32)
    ks, vs = (1.0, 1.0) if kv_cache_dtype == "auto" else (0.5, 2.0)
    # fill the caches through the schema-level cache write, token by token batch (slots of every position)
    for pos in range(max_len):
        live = [i for i in range(S) if pos < seq_lens[i]]
        if pos % 97 and pos > 3 and pos < max_len - 2:      # keep the oracle loop short: write most positions in bulk below
            continue
        k = (rng.standard_normal((len(live), Hkv, D)) * 0.3).astype(np.float16)
        v = (rng.standard_normal((len(live), Hkv, D)) * 0.3).astype(np.float16)
        slots = np.array([bt[i, pos // BS] * BS + pos % BS for i in live], np.int64)
        T.cache.reshape_and_cache(t(k), t(v), kc, vc, t(slots), kv_cache_dtype, ks, vs)
        oa.reshape_and_cache(k, v, kc_ref, vc_ref, slots, kv_cache_dtype, ks, vs)
    np.testing.assert_array_equal(kc.cpu().numpy().view(np.uint8), kc_ref.view(np.uint8))
    np.testing.assert_array_equal(vc.cpu().numpy().view(np.uint8), vc_ref.view(np.uint8))
    q = t(rng.standard_normal((S, Hq, D)).astype(np.float32), dtype)
    scale = float(D ** -0.5)
    ref = oa.paged_attention_decode(q.float().cpu().numpy(), kc_ref if kv_cache_dtype != "auto" else kc_ref.astype(np.float32),
                                    vc_ref if kv_cache_dtype != "auto" else vc_ref.astype(np.float32), bt, seq_lens, scale,
                                    None, kv_cache_dtype, ks, vs)
    atol = 1e-3 if kv_cache_dtype == "auto" else 1e-2
    common = (q, kc, vc, Hkv, scale, t(bt), t(seq_lens), BS, max_len, None, kv_cache_dtype, ks, vs)
    out1 = torch.empty(S, Hq, D, dtype=dtype, device=DEV)
    captured(lambda: T.C.paged_attention_v1(out1, *common, 0, 0, 0, 64, 0))
    np.testing.assert_allclose(out1.float().cpu().numpy(), ref, atol=atol, rtol=1e-2)
    P = (max_len + 511) // 512
    es = torch.empty(S, Hq, P, device=DEV)
    ml = torch.empty(S, Hq, P, device=DEV)
    tmp = torch.empty(S, Hq, P, D, dtype=dtype, device=DEV)
    out2 = torch.empty_like(out1)
    captured(lambda: T.C.paged_attention_v2(out2, es, ml, tmp, *common, 0, 0, 0, 64, 0))
    np.testing.assert_allclose(out2.float().cpu().numpy(), ref, atol=atol, rtol=1e-2)
    out3 = torch.empty_like(out1)
    captured(lambda: T.rocm.paged_attention(out3, es, ml, tmp, *common))
    np.testing.assert_allclose(out3.float().cpu().numpy(), ref, atol=atol, rtol=1e-2)


def test_schema_fp8_quant_ops_write_in_place(T):
    rng = np.random.default_rng(3)
    M, K = 33, 1024
    x = t(((rng.random((M, K)) - 0.5) * 60).astype(np.float32), torch.bfloat16)
    xr = x.float().cpu().numpy()
    # static
    out = torch.empty(M, K, dtype=torch.float8_e4m3fn, device=DEV)
    ptr = out.data_ptr()
    sc = torch.tensor([0.37], device=DEV)
    captured(lambda: T.C.static_scaled_fp8_quant(out, x, sc))
    assert out.data_ptr() == ptr
    np.testing.assert_array_equal(out.view(torch.uint8).cpu().numpy(), of8.static_scaled_fp8_quant(xr, np.float32(0.37)))
    # dynamic per tensor (Tensor! scale)
    out.zero_()
    scale = torch.zeros(1, device=DEV)
    captured(lambda: T.C.dynamic_scaled_fp8_quant(out, x, scale))
    rq, rs = of8.dynamic_scaled_fp8_quant(xr)
    np.testing.assert_array_equal(scale.cpu().numpy(), rs)
    np.testing.assert_array_equal(out.view(torch.uint8).cpu().numpy(), rq)
    # dynamic per token, out padded to max(M, 17) rows like the reference allocates it (_custom_ops.py:661-664)
    outp = torch.zeros(M + 4, K, dtype=torch.float8_e4m3fn, device=DEV)
    scales = torch.zeros(M + 4, 1, device=DEV)
    captured(lambda: T.C.dynamic_per_token_scaled_fp8_quant(outp, x, scales, None))
    rq, rs = of8.dynamic_per_token_scaled_fp8_quant(xr)
    np.testing.assert_array_equal(outp.view(torch.uint8).cpu().numpy()[:M], rq)
    np.testing.assert_array_equal(scales.cpu().numpy()[:M], rs)
    assert (outp.view(torch.uint8)[M:] == 0).all()
    # MI300's fnuz encoding is refused, not silently mis-scaled by 2x
    if hasattr(torch, "float8_e4m3fnuz"):
        bad = torch.empty(M, K, dtype=torch.float8_e4m3fnuz, device=DEV)
        with pytest.raises(RuntimeError, match="fnuz"):
            T.C.static_scaled_fp8_quant(bad, x, sc)


@pytest.mark.parametrize("M", [8, 200])
def test_schema_cutlass_scaled_mm(T, M):
    rng = np.random.default_rng(4 + M)
    K, N = 1024, 256
    a = t((rng.standard_normal((M, K)) * 2).astype(np.float32)).to(torch.float8_e4m3fn)
    w = t((rng.standard_normal((N, K)) * 2).astype(np.float32)).to(torch.float8_e4m3fn)
    sa = t((rng.random((M, 1)) * 0.02 + 0.002).astype(np.float32))
    sb = t((rng.random((N, 1)) * 0.02 + 0.002).astype(np.float32))
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ptr = out.data_ptr()
    if M <= 64:
        captured(lambda: T.C.cutlass_scaled_mm(out, a, w.t(), sa, sb, None))
    else:
        T.C.cutlass_scaled_mm(out, a, w.t(), sa, sb, None)
    assert out.data_ptr() == ptr
    ref = of8.scaled_mm(a.view(torch.uint8).cpu().numpy(), w.view(torch.uint8).cpu().numpy().T, sa.cpu().numpy(),
                        sb.cpu().numpy().reshape(-1), None)
    np.testing.assert_allclose(out.float().cpu().numpy(), ref, rtol=1.6e-2, atol=1.6e-2 * np.abs(ref).max())
    assert T.C.cutlass_scaled_mm_supports_fp8(95) is True


def _qtype(bits, bias):
    """The b_q_type argument: the torchbind class _core_C.ScalarType (kernels/core/torch_bindings.cpp:10-13; standalone this
    package registers it, csrc_torch/core_scalar_type.cpp) -- or the size in bits where no library has the class."""
    from aphrodite_engine_amd import torch_cpp
    cls = torch_cpp.ensure_scalar_type_class()
    return bits if cls is None else cls.uint(bits, bias)


def test_core_scalar_type_class():
    """_core_C.ScalarType as this package registers it: the reference class's Python surface (kernels/core/scalar_type.hpp;
    the typing mock of aphrodite/_core_ext.py:28-171 lists it) and its values for the types the hot path and the reference's
    scalar_types table name (aphrodite/scalar_type.py)."""
    from aphrodite_engine_amd import torch_cpp
    S = torch_cpp.ensure_scalar_type_class()
    assert S is not None
    u4b8, u4, i4, u8b128 = S.uint(4, 8), S.uint(4, None), S.int_(4, None), S.uint(8, 128)
    assert (str(u4b8), str(u4), str(i4), str(u8b128)) == ("uint4b8", "uint4", "int4", "uint8b128")
    assert repr(i4) == "ScalarType.int4"
    assert (u4b8.size_bits, u4b8.bias, u4b8.mantissa, u4b8.exponent, u4b8.signed) == (4, 8, 4, 0, False)
    assert (u4b8.min(), u4b8.max(), u4.min(), u4.max(), i4.min(), i4.max(), u8b128.min(), u8b128.max()) == (-8, 7, 0, 15, -8, 7, -128, 127)
    assert u4b8.is_integer() and not u4b8.is_floating_point() and u4b8.has_bias() and not u4.has_bias() and i4.is_signed()
    assert u4b8 == S.uint(4, 8) and not (u4b8 == u4) and S(0, 4, 8, False) == u4b8
    e4m3fn, e5m2, f16, bf16, e3m2f = S.float_(4, 3, True, 2), S.float_IEEE754(5, 2), S.float_IEEE754(5, 10), S.float_IEEE754(8, 7), S.float_(3, 2, True, 0)
    assert [str(x) for x in (e4m3fn, e5m2, f16, bf16, e3m2f)] == ["float8_e4m3fn", "float8_e5m2", "float16_e5m10", "float16_e8m7", "float6_e3m2f"]
    assert (e4m3fn.max(), e4m3fn.min(), e5m2.max(), f16.max(), e3m2f.max()) == (448.0, -448.0, 57344.0, 65504.0, 28.0)
    assert bf16.max() == float(torch.finfo(torch.bfloat16).max)
    assert e4m3fn.has_nans() and not e4m3fn.has_infs() and not e4m3fn.is_ieee_754() and f16.is_ieee_754() and f16.has_infs()
    assert not e3m2f.has_nans() and e4m3fn.is_floating_point() and e4m3fn.is_signed()
    for x in (u4b8, u8b128, i4, e4m3fn, bf16):                       # torch.compile's flatten / unflatten round trip
        assert S.__obj_unflatten__(x.__obj_flatten__()) == x
    with pytest.raises(TypeError):
        len(u4b8)
    with pytest.raises(RuntimeError):
        S.float_(4, 3, False, 1)                                       # IEEE types go through float_IEEE754


def test_schema_gptq_marlin_gemm(T):
    """_C::gptq_marlin_gemm (torch_bindings.cpp:195-201) with the verbatim schema -- b_q_type is the torchbind class
    _core_C.ScalarType; symmetric uint4b8 weights, no per-call allocation under capture; a type outside the Marlin role's
    table is refused."""
    rng = np.random.default_rng(5)
    K, N, G, M = 1024, 256, 128, 16
    w = (rng.standard_normal((K, N)) * 0.02).astype(np.float16)
    w_ref, q, s, _ = oq.quantize_weights(w, 4, G)                       # symmetric: stored q in [0, 15], zero point 8
    qw = T.C.gptq_marlin_repack(t(oq.gptq_pack(q)), torch.empty(0, dtype=torch.int32, device=DEV), K, N, 4)
    a = rng.standard_normal((M, K)).astype(np.float16)
    empty = torch.empty(0, dtype=torch.int32, device=DEV)
    ws = torch.zeros(N // 64 * 16, dtype=torch.int32, device=DEV)
    args = (t(a), qw, t(s.astype(np.float16)), empty, empty, empty, ws, _qtype(4, 8), M, N, K, True, False, True, False)
    assert "ScalarType b_q_type" in str(T.C.gptq_marlin_gemm.default._schema)
    _, out = captured(lambda: T.C.gptq_marlin_gemm(*args))
    ref = a.astype(np.float64) @ w_ref.astype(np.float64)
    got = out.float().cpu().numpy()
    assert np.abs(got - ref).mean() / np.abs(ref).mean() < 0.04            # tests/kernels/test_marlin_gemm.py:57-59
    np.testing.assert_allclose(got, ref, rtol=2e-3, atol=2e-3 * np.abs(ref).max())


def test_cpp_registered_ops_match_the_python_registration(T):
    """The ops registered from C++ (csrc_torch/torch_bindings.cpp, INTEGRATION.md option 3) give the same bits as the
    Python-registered ones (same C ABI underneath), decode- and prefill-sized, also under HIP-graph capture."""
    from aphrodite_engine_amd import torch_cpp
    torch_cpp.load()
    C, cache = torch.ops._C_mi355x, torch.ops._C_mi355x_cache_ops
    rng = np.random.default_rng(11)
    # gptq_gemm: M = 32 (decode kernel), 300 (prefill-sized MFMA kernel), act-order at 300
    K, N, G = 1024, 512, 128
    qw, qz, s_ = make_gptq(rng, K, N, G)
    shuf = t(qw.copy())
    empty = torch.empty(0, dtype=torch.int32, device=DEV)
    T.C.gptq_shuffle(shuf, empty, 4)
    qz_d, s_d = t(qz), t(s_)
    for M in (32, 300):
        a = t(rng.standard_normal((M, K)).astype(np.float16))
        want = T.C.gptq_gemm(a, shuf, qz_d, s_d, empty, True, 4)
        _, got = captured(lambda: C.gptq_gemm(a, shuf, qz_d, s_d, empty, True, 4))
        assert torch.equal(got, want)
    # 8-bit weights through the same op (csrc/wnx_gemm.hip): the small-M kernel and the reconstruct + GEMM rule
    from oracle import quant as oq
    q8 = rng.integers(0, 256, size=(K, N))
    z8 = rng.integers(0, 256, size=(K // G, N))
    qw8, qz8 = t(oq.gptq_pack(q8, 8)), t(oq.pack_cols(z8, 8))
    for M in (8, 100):
        a = t(rng.standard_normal((M, K)).astype(np.float16))
        want = T.C.gptq_gemm(a, qw8, qz8, s_d, empty, True, 8)
        got = C.gptq_gemm(a, qw8, qz8, s_d, empty, True, 8)
        assert torch.equal(got, want)
    # cutlass_scaled_mm: M = 40 and 200
    for M in (40, 200):
        K2, N2 = 512, 256
        a8 = t((rng.standard_normal((M, K2)) * 2).astype(np.float32)).to(torch.float8_e4m3fn)
        w8 = t((rng.standard_normal((N2, K2)) * 2).astype(np.float32)).to(torch.float8_e4m3fn)
        sa = t((rng.random((M, 1)) * 0.1 + 0.01).astype(np.float32))
        sb = t((rng.random((N2, 1)) * 0.1 + 0.01).astype(np.float32))
        bias = t(rng.standard_normal(N2).astype(np.float32), torch.bfloat16)
        want = torch.empty(M, N2, dtype=torch.bfloat16, device=DEV)
        got = torch.empty_like(want)
        T.C.cutlass_scaled_mm(want, a8, w8.t(), sa, sb, bias)
        C.cutlass_scaled_mm(got, a8, w8.t(), sa, sb, bias)
        assert torch.equal(got, want)
    # cache write + decode attention
    S, Hq, Hkv, D, BS = 3, 8, 2, 128, 16
    seq_lens = np.array([5, 130, 600], np.int32)
    bps = (int(seq_lens.max()) + BS - 1) // BS
    NB = S * bps
    kc = (torch.randn(NB, Hkv, D // 8, BS, 8, device=DEV) * 0.3).half()
    vc = (torch.randn(NB, Hkv, D, BS, device=DEV) * 0.3).half()
    kc2, vc2 = kc.clone(), vc.clone()
    bt = rng.permutation(NB).reshape(S, bps).astype(np.int32)
    k = t((rng.standard_normal((S, Hkv, D)) * 0.3).astype(np.float16))
    v = t((rng.standard_normal((S, Hkv, D)) * 0.3).astype(np.float16))
    slots = t(np.array([bt[i, (seq_lens[i] - 1) // BS] * BS + (seq_lens[i] - 1) % BS for i in range(S)], np.int64))
    T.cache.reshape_and_cache(k, v, kc, vc, slots, "auto", 1.0, 1.0)
    cache.reshape_and_cache(k, v, kc2, vc2, slots, "auto", 1.0, 1.0)
    assert torch.equal(kc, kc2) and torch.equal(vc, vc2)
    q = t(rng.standard_normal((S, Hq, D)).astype(np.float32), torch.float16)
    common = (q, kc, vc, Hkv, float(D ** -0.5), t(bt), t(seq_lens), BS, int(seq_lens.max()), None, "auto", 1.0, 1.0, 0, 0, 0, 64, 0)
    want = torch.empty(S, Hq, D, dtype=torch.float16, device=DEV)
    got = torch.empty_like(want)
    T.C.paged_attention_v1(want, *common)
    captured(lambda: C.paged_attention_v1(got, *common))
    assert torch.equal(got, want)


def test_cpp_registered_memory_bound_and_cache_ops_match_the_python_registration(T):
    """Round 4: twelve more ops registered from C++ (norms, SiluAndMul, rotary, partitioned attention, gptq_shuffle,
    awq_dequantize, two FP8 quantisers, advance_step, reshape_and_cache_flash, convert_fp8) -- same C ABI underneath, so the
    same bits as the Python-registered ops (which the oracle tests check)."""
    from aphrodite_engine_amd import torch_cpp
    torch_cpp.load()
    C, cache = torch.ops._C_mi355x, torch.ops._C_mi355x_cache_ops
    rng = np.random.default_rng(12)
    for dtype in (torch.float16, torch.bfloat16):
        x = t(rng.standard_normal((7, 1024)).astype(np.float32), dtype)
        w = t((1 + 0.1 * rng.standard_normal(1024)).astype(np.float32), dtype)
        a, b = torch.empty_like(x), torch.empty_like(x)
        T.C.rms_norm(a, x, w, 1e-5)
        C.rms_norm(b, x, w, 1e-5)
        assert torch.equal(a, b)
        r = t(rng.standard_normal((7, 1024)).astype(np.float32), dtype)
        x1, r1, x2, r2 = x.clone(), r.clone(), x.clone(), r.clone()
        T.C.fused_add_rms_norm(x1, r1, w, 1e-5)
        C.fused_add_rms_norm(x2, r2, w, 1e-5)
        assert torch.equal(x1, x2) and torch.equal(r1, r2)
        g = t(rng.standard_normal((7, 2048)).astype(np.float32), dtype)
        a, b = torch.empty(7, 1024, dtype=dtype, device=DEV), torch.empty(7, 1024, dtype=dtype, device=DEV)
        T.C.silu_and_mul(a, g)
        C.silu_and_mul(b, g)
        assert torch.equal(a, b)
        # rotary (neox and gptj), in place
        from aphrodite_engine_amd.model import _rope_cache
        cs = _rope_cache(128, 256, 10000.0, dtype, DEV)
        pos = t(rng.integers(0, 256, size=7).astype(np.int64))
        for neox in (True, False):
            q = t(rng.standard_normal((7, 8 * 128)).astype(np.float32), dtype)
            k = t(rng.standard_normal((7, 2 * 128)).astype(np.float32), dtype)
            q2, k2 = q.clone(), k.clone()
            T.C.rotary_embedding(pos, q, k, 128, cs, neox)
            C.rotary_embedding(pos, q2, k2, 128, cs, neox)
            assert torch.equal(q, q2) and torch.equal(k, k2)
        # FP8 quantisers
        xq = t(rng.standard_normal((9, 512)).astype(np.float32), dtype)
        sc = t(np.array([0.07], np.float32))
        a, b = (torch.empty(9, 512, dtype=torch.float8_e4m3fn, device=DEV) for _ in range(2))
        T.C.static_scaled_fp8_quant(a, xq, sc)
        C.static_scaled_fp8_quant(b, xq, sc)
        assert torch.equal(a.view(torch.uint8), b.view(torch.uint8))
        s1, s2 = (torch.empty(9, 1, dtype=torch.float32, device=DEV) for _ in range(2))
        T.C.dynamic_per_token_scaled_fp8_quant(a, xq, s1, None)
        C.dynamic_per_token_scaled_fp8_quant(b, xq, s2, None)
        assert torch.equal(a.view(torch.uint8), b.view(torch.uint8)) and torch.equal(s1, s2)
    # gptq_shuffle (with and without act-order) and awq_dequantize
    K, N, G = 1024, 256, 128
    qw, qz, s_ = make_gptq(rng, K, N, G)
    perm = t(rng.permutation(K).astype(np.int32))
    empty = torch.empty(0, dtype=torch.int32, device=DEV)
    for p_ in (empty, perm):
        a, b = t(qw.copy()), t(qw.copy())
        T.C.gptq_shuffle(a, p_, 4)
        C.gptq_shuffle(b, p_, 4)
        assert torch.equal(a, b)
    aw = t(rng.integers(-2 ** 31, 2 ** 31 - 1, size=(K, N // 8), dtype=np.int64).astype(np.int32))
    az = t(rng.integers(-2 ** 31, 2 ** 31 - 1, size=(K // G, N // 8), dtype=np.int64).astype(np.int32))
    asc = t((rng.random((K // G, N)) * 0.01).astype(np.float16))
    assert torch.equal(T.C.awq_dequantize(aw, asc, az, 0, 0, 0), C.awq_dequantize(aw, asc, az, 0, 0, 0))
    # partitioned decode attention (v2) and the flash-layout cache write, convert_fp8
    S, Hq, Hkv, D, BS = 3, 8, 2, 128, 16
    seq_lens = np.array([5, 700, 1300], np.int32)
    bps = (int(seq_lens.max()) + BS - 1) // BS
    NB = S * bps
    kc = (torch.randn(NB, Hkv, D // 8, BS, 8, device=DEV) * 0.3).half()
    vc = (torch.randn(NB, Hkv, D, BS, device=DEV) * 0.3).half()
    bt = rng.permutation(NB).reshape(S, bps).astype(np.int32)
    q = t(rng.standard_normal((S, Hq, D)).astype(np.float32), torch.float16)
    nparts = (int(seq_lens.max()) + 511) // 512
    outs = []
    for ns in (T.C, C):
        out = torch.empty(S, Hq, D, dtype=torch.float16, device=DEV)
        es = torch.empty(S, Hq, nparts, dtype=torch.float32, device=DEV)
        ml = torch.empty_like(es)
        tmp = torch.empty(S, Hq, nparts, D, dtype=torch.float16, device=DEV)
        ns.paged_attention_v2(out, es, ml, tmp, q, kc, vc, Hkv, float(D ** -0.5), t(bt), t(seq_lens), BS, int(seq_lens.max()), None,
                              "auto", 1.0, 1.0, 0, 0, 0, 64, 0)
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    kf = torch.zeros(NB, BS, Hkv, D, dtype=torch.float16, device=DEV)
    vf = torch.zeros_like(kf)
    kf2, vf2 = kf.clone(), vf.clone()
    k = t((rng.standard_normal((S, Hkv, D)) * 0.3).astype(np.float16))
    v = t((rng.standard_normal((S, Hkv, D)) * 0.3).astype(np.float16))
    slots = t(np.array([3, 77, 200], np.int64))
    T.cache.reshape_and_cache_flash(k, v, kf, vf, slots, "auto", 1.0, 1.0)
    cache.reshape_and_cache_flash(k, v, kf2, vf2, slots, "auto", 1.0, 1.0)
    assert torch.equal(kf, kf2) and torch.equal(vf, vf2) and float(kf.abs().sum()) > 0
    src = t((rng.standard_normal((64, 128)) * 0.5).astype(np.float16))
    d1, d2 = (torch.empty(64, 128, dtype=torch.uint8, device=DEV) for _ in range(2))
    T.cache.convert_fp8(d1, src, 0.5, "fp8")
    cache.convert_fp8(d2, src, 0.5, "fp8")
    assert torch.equal(d1, d2)
    # advance_step
    nq = 4
    mk = lambda: (t(np.arange(nq, dtype=np.int64)), t(np.array([11, 12, 13, 14], np.int64)), t(np.array([5, 130, 600, 15], np.int64)),
                  t(np.array([6, 131, 601, 16], np.int32)), t(np.zeros(nq, np.int64)),
                  t(rng.permutation(nq * 40).reshape(nq, 40).astype(np.int32)))
    a_, b_ = mk(), mk()
    b_ = tuple(x.clone() for x in a_)
    T.C.advance_step_flashattn(nq, nq, BS, *a_)
    C.advance_step_flashattn(nq, nq, BS, *b_)
    for x, y in zip(a_, b_):
        assert torch.equal(x, y)


def test_reference_wrapper_call_list_resolves_on_the_device_box(T):
    """tests/golden/ref_custom_ops_calls.json = every ``torch.ops.<ns>.<op>(...)`` call the reference's own
    ``aphrodite/_custom_ops.py`` wrappers make for the hot-path ops, with the number of positional arguments they pass
    (generated here from /root/reference by tests/test_reference_binding_cpu.py, which also checks it is current).  On
    the GPU box the reference is absent: check the committed list against the ops actually registered -- the op exists
    in the mirrored namespace and its schema takes exactly that many arguments."""
    import json
    import os
    calls = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_custom_ops_calls.json")))
    ns_map = {"_C": "_aphro_g_C", "_C_cache_ops": "_aphro_g_cache", "_rocm_C": "_aphro_g_rocm", "_moe_C": "_aphro_g_moe"}
    assert len(calls) >= 20
    for wrapper, cl in calls.items():
        for ns, op, nargs in cl:
            packet = getattr(getattr(torch.ops, ns_map[ns]), op)        # AttributeError = the reference's hint_on_error case
            schema = packet.default._schema
            assert len(schema.arguments) == nargs, (wrapper, op, str(schema))


def test_cpp_registered_round5_ops_match_the_python_registration(T):
    """Round 5: dynamic_scaled_fp8_quant, moe_align_block_size, _moe::topk_softmax, swap_blocks, copy_blocks from C++ -- the
    same C ABI underneath, the same results as the Python-registered ops."""
    from aphrodite_engine_amd import torch_cpp
    torch_cpp.load()
    C, cache, moe = torch.ops._C_mi355x, torch.ops._C_mi355x_cache_ops, torch.ops._C_mi355x_moe
    rng = np.random.default_rng(3)
    x = t(rng.standard_normal((9, 512)).astype(np.float32), torch.float16)
    a, b = (torch.empty(9, 512, dtype=torch.float8_e4m3fn, device=DEV) for _ in range(2))
    sa, sb = torch.zeros(1, device=DEV), torch.full((1, ), 123.0, device=DEV)      # (C++ zeroes the scale itself)
    T.C.dynamic_scaled_fp8_quant(a, x, sa)
    C.dynamic_scaled_fp8_quant(b, x, sb)
    assert torch.equal(a.view(torch.uint8), b.view(torch.uint8)) and torch.equal(sa, sb)
    # routing
    gating = t(rng.standard_normal((33, 8)).astype(np.float32))
    outs = []
    for ns in (T.moe, moe):
        w = torch.empty(33, 2, device=DEV)
        ids = torch.empty(33, 2, dtype=torch.int32, device=DEV)
        src = torch.empty(33, 2, dtype=torch.int32, device=DEV)
        ns.topk_softmax(w, ids, src, gating)
        outs.append((w, ids, src))
    assert all(torch.equal(p, q) for p, q in zip(*outs))
    ids = outs[0][1]
    res = []
    for ns in (T.C, C):
        cap = ids.numel() + 8 * 15
        sorted_ids = torch.full((cap, ), -1, dtype=torch.int32, device=DEV)
        experts = torch.full(((cap + 15) // 16, ), -1, dtype=torch.int32, device=DEV)
        post = torch.zeros(1, dtype=torch.int32, device=DEV)
        ns.moe_align_block_size(ids, 8, 16, sorted_ids, experts, post)
        res.append((sorted_ids, experts, post))
    n = int(res[0][2].item())
    assert int(res[1][2].item()) == n and torch.equal(res[0][0][:n], res[1][0][:n]) and torch.equal(res[0][1][:n // 16], res[1][1][:n // 16])
    # block copies: two layers, pairs (0 -> 3), (2 -> 1)
    kc = [t(rng.standard_normal((4, 2, 16, 16, 8)).astype(np.float32), torch.float16) for _ in range(2)]
    vc = [t(rng.standard_normal((4, 2, 128, 16)).astype(np.float32), torch.float16) for _ in range(2)]
    kc2, vc2 = [c.clone() for c in kc], [c.clone() for c in vc]
    bm = torch.tensor([[0, 3], [2, 1]], dtype=torch.int64, device=DEV)
    T.cache.copy_blocks(kc, vc, bm)
    cache.copy_blocks(kc2, vc2, bm)
    assert all(torch.equal(p, q) for p, q in zip(kc + vc, kc2 + vc2)) and torch.equal(kc2[1][3], kc2[1][0])
    src_cache = t(rng.standard_normal((4, 2, 128, 16)).astype(np.float32), torch.float16)
    d1, d2 = torch.zeros_like(src_cache), torch.zeros_like(src_cache)
    bmh = torch.tensor([[1, 0], [3, 2]], dtype=torch.int64)
    T.cache.swap_blocks(src_cache, d1, bmh)
    cache.swap_blocks(src_cache, d2, bmh)
    torch.cuda.synchronize()
    assert torch.equal(d1, d2) and torch.equal(d2[0], src_cache[1]) and torch.equal(d2[2], src_cache[3])
    host = torch.zeros(4, 2, 128, 16, dtype=torch.float16).pin_memory()
    cache.swap_blocks(src_cache, host, bmh)                  # device -> host
    torch.cuda.synchronize()
    assert torch.equal(host[0], src_cache[1].cpu())


def test_cpp_registered_round5_second_batch_matches_the_python_registration(T):
    """Round 5, second batch from C++: awq_gemm (decode-sized and prompt-sized M), cutlass_scaled_mm_supports_fp8,
    _rocm_C::paged_attention (whole-sequence and partitioned forms), _C_custom_ar::all_reduce_reg / all_reduce_unreg /
    meta_size (on a loopback communicator: the kernel sums `world` copies of the local buffer) -- the same C ABI underneath,
    the same bits as the Python-registered ops."""
    from aphrodite_engine_amd import _custom_ops as ops, torch_cpp
    from aphrodite_engine_amd.distributed.custom_all_reduce import LoopbackAllreduce
    torch_cpp.load()
    C, rocm, car = torch.ops._C_mi355x, torch.ops._rocm_C_mi355x, torch.ops._C_mi355x_custom_ar
    assert C.cutlass_scaled_mm_supports_fp8(94) is True
    rng = np.random.default_rng(11)
    K, N, G = 512, 256, 128
    w = (rng.standard_normal((K, N)) * 0.02).astype(np.float16)
    _, q, s, zp = oq.quantize_weights(w, 4, G, zero_points=True)
    qw, qz, sc = t(oq.awq_pack(q)), t(oq.awq_pack(zp)), t(s.astype(np.float16))
    for M in (1, 33, 70, 300):                       # one pass, two 64-row passes, the repack + tile-machine path
        a = t(rng.standard_normal((M, K)).astype(np.float16))
        assert torch.equal(C.awq_gemm(a, qw, sc, qz, 8), T.C.awq_gemm(a, qw, sc, qz, 8)), M
    # the Marlin-role prepacks (load time)
    assert torch.equal(C.awq_marlin_repack(qw, K, N, 4), T.C.awq_marlin_repack(qw, K, N, 4))
    gq = t(oq.gptq_pack(q))
    perm = t(rng.permutation(K).astype(np.int32))
    for pm in (torch.empty(0, dtype=torch.int32, device=DEV), perm):
        assert torch.equal(C.gptq_marlin_repack(gq, pm, K, N, 4), T.C.gptq_marlin_repack(gq, pm, K, N, 4))
    # the Marlin-role GEMMs: symmetric uint4b8 and zero-point weights, decode-sized / one-pass 33..64-row / prompt-sized rows
    K2, N2 = 4096, 8192                               # n k = 2^25: the size from which the one-pass kernel is preferred
    w2 = (rng.standard_normal((K2, N2)) * 0.02).astype(np.float16)
    _, q2, s2, zp2 = oq.quantize_weights(w2, 4, G, zero_points=True)
    empty = torch.empty(0, dtype=torch.int32, device=DEV)
    mq = T.C.gptq_marlin_repack(t(oq.gptq_pack(q2)), empty, K2, N2, 4)
    msc, mzp = t(s2.astype(np.float16)), t(oq.gptq_pack_zeros(zp2 + 1))        # plain zero points (no GPTQ "- 1")
    ws_m = torch.zeros(N2 // 64 * 16, dtype=torch.int32, device=DEV)
    for M in (7, 48, 130):
        a2 = t(rng.standard_normal((M, K2)).astype(np.float16))
        for has_zp, zt in ((False, empty), (True, mzp)):
            args = (a2, mq, msc, zt, empty, empty, ws_m, _qtype(4, 0 if has_zp else 8), M, N2, K2, True, has_zp, True, False)
            assert torch.equal(C.gptq_marlin_gemm(*args), T.C.gptq_marlin_gemm(*args)), (M, has_zp)
    assert "ScalarType b_q_type" in str(C.gptq_marlin_gemm.default._schema)        # the C++ registration: verbatim schema, boxed kernel
    with pytest.raises(RuntimeError):                                              # int3 is not in the role's table
        C.gptq_marlin_gemm(a2, mq, msc, empty, empty, empty, ws_m, _qtype(3, 4), 130, N2, K2, True, False, True, False)
    # ... and the role's 8-bit type (uint8b128): repack (act-order rows made sequential) and GEMM, C++ == Python registration
    _, q8, s8, _ = oq.quantize_weights((rng.standard_normal((512, 256)) * 0.05).astype(np.float32), 8, G, zero_points=False)
    gq8 = t(oq.gptq_pack(q8, 8).astype(np.int32))
    perm8 = t(rng.permutation(512).astype(np.int32))
    for pm in (empty, perm8):
        r8 = C.gptq_marlin_repack(gq8, pm, 512, 256, 8)
        assert torch.equal(r8, T.C.gptq_marlin_repack(gq8, pm, 512, 256, 8))
        for M in (7, 48, 130):
            a8b = t(rng.standard_normal((M, 512)).astype(np.float16))
            args = (a8b, r8, t(s8.astype(np.float16)), empty, empty, pm, ws_m, _qtype(8, 128), M, 256, 512, True, False, True, False)
            assert torch.equal(C.gptq_marlin_gemm(*args), T.C.gptq_marlin_gemm(*args)), (M, pm.numel())
    w8 = t((rng.standard_normal((256, 512)) * 0.5).astype(np.float32)).to(torch.float8_e4m3fn)     # [N, K]
    for sb8 in (t(np.array([0.02], np.float32)), t((rng.random(256) * 0.02 + 0.01).astype(np.float32))):
        for M in (5, 70, 200):
            a8 = t(rng.standard_normal((M, 512)).astype(np.float16))
            args = (a8, w8, sb8, ws_m, 8, M, 256, 512)
            assert torch.equal(C.fp8_marlin_gemm(*args), T.C.fp8_marlin_gemm(*args)), M
    a = t(rng.standard_normal((16, 2 * K)).astype(np.float16))[:, ::2]       # non-unit column stride: made contiguous
    assert torch.equal(C.awq_gemm(a, qw, sc, qz, 8), T.C.awq_gemm(a, qw, sc, qz, 8))
    # paged attention through the ROCm schema
    S, Hq, Hkv, D, BS = 5, 8, 2, 128, 16
    for seq_lens in (np.array([1, 40, 300, 511, 77], np.int32), np.array([5, 600, 1300, 33, 2], np.int32)):
        max_len = int(seq_lens.max())
        bps = (max_len + BS - 1) // BS
        NB = S * bps + 1
        kc = t((rng.standard_normal((NB, Hkv, D // 8, BS, 8)) * 0.3).astype(np.float16))
        vc = t((rng.standard_normal((NB, Hkv, D, BS)) * 0.3).astype(np.float16))
        bt = t(rng.permutation(NB)[:S * bps].reshape(S, bps).astype(np.int32))
        qq = t(rng.standard_normal((S, Hq, D)).astype(np.float16))
        P = (max_len + 511) // 512
        outs = []
        for ns in (T.rocm, rocm):
            es, ml = torch.zeros(S, Hq, P, device=DEV), torch.zeros(S, Hq, P, device=DEV)
            tmp = torch.zeros(S, Hq, P, D, dtype=torch.float16, device=DEV)
            out = torch.empty(S, Hq, D, dtype=torch.float16, device=DEV)
            ns.paged_attention(out, es, ml, tmp, qq, kc, vc, Hkv, float(D ** -0.5), bt, t(seq_lens), BS, max_len, None, "auto", 1.0, 1.0)
            outs.append((out, es, ml, tmp))
        assert all(torch.equal(p, q2) for p, q2 in zip(*outs)), seq_lens
    # the peer-access all-reduce on a loopback communicator of 4 "ranks"
    assert car.meta_size() == ops.meta_size()
    lb = LoopbackAllreduce(4, torch.device(DEV), max_size=1 << 20)
    try:
        for dtype, numel in ((torch.float16, 32 * 4096), (torch.bfloat16, 64 * 4096), (torch.float32, 8 * 1024)):
            x = t(rng.standard_normal(numel).astype(np.float32), dtype)
            want = torch.empty_like(x)
            ops.all_reduce_reg(lb._ptr, x, want)
            got = torch.empty_like(x)
            car.all_reduce_reg(lb._ptr, x, got)
            torch.cuda.synchronize()
            assert torch.equal(got, want)
            reg = torch.empty(1 << 20, dtype=torch.uint8, device=DEV)
            got2 = torch.empty_like(x)
            car.all_reduce_unreg(lb._ptr, x, reg, got2)
            torch.cuda.synchronize()
            assert torch.equal(got2, want)
        assert not ops.custom_ar_error(lb._ptr)
        with pytest.raises(RuntimeError):
            car.all_reduce_unreg(lb._ptr, torch.zeros(1 << 20, dtype=torch.float16, device=DEV),
                                 torch.empty(16, dtype=torch.uint8, device=DEV), torch.zeros(1 << 20, dtype=torch.float16, device=DEV))
    finally:
        lb.close()
