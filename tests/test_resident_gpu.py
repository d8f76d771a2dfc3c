"""The resident-activation decode GEMM (csrc/wna16_gemm_resident.hip, round 3) at the BASELINE configs[1] shapes
against the ORACLE (oracle.quant, pinned by the reference's own CUDA kernels run on a host shim --
tests/test_oracle_golden.py::test_gptq_*reference*), in every output form the decode step launches, both weight layouts,
f16 and bf16, M = 1 .. 32; plus bit-equality with the round-2 kernel where the K partition is the same (that kernel is
itself oracle-checked in tests/test_headline_gpu.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import attention as oa
from oracle import quant as oq
from tests.test_headline_gpu import SHAPES, case, t, unpack_a

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X (torch.cuda.is_available() is False)")
    from aphrodite_engine_amd import _custom_ops, _lib
    _lib.lib()
    return _custom_ops


@pytest.fixture(autouse=True)
def _no_cfg_env(ops):
    os.environ.pop("APHRO_WNA16_RES_CFG", None); ops.reload_env()
    yield
    os.environ.pop("APHRO_WNA16_RES_CFG", None); ops.reload_env()


@pytest.mark.parametrize("M", [1, 8, 16, 17, 32])
@pytest.mark.parametrize("K,N", SHAPES)
def test_resident_slabs_and_out_vs_oracle(ops, K, N, M):
    shuf, qzeros, scales, a, ref = case(K, N)
    ref = ref[:M]
    G = K // 128
    ks = ops.wna16_resident_ksplit(M, N, K, G)
    assert ks >= 1, "every configs[1] projection is served by the resident kernel"
    pk = ops.wna16_pack_a(t(a[:M]))
    slabs, ks2 = ops.wna16_gemm_resident(pk, M, K, t(shuf), t(qzeros), t(scales), 1, mode="slabs")
    assert ks2 == ks and slabs.shape == (ks, M, N)
    s = slabs.double().sum(0).cpu().numpy()
    np.testing.assert_allclose(s, ref, rtol=1e-4, atol=2e-5 * np.abs(ref).max())     # fp32 accumulate vs fp64 oracle
    # deterministic (no atomics)
    slabs_b, _ = ops.wna16_gemm_resident(pk, M, K, t(shuf), t(qzeros), t(scales), 1, mode="slabs")
    assert torch.equal(slabs, slabs_b)
    if ks == 1:
        y = ops.wna16_gemm_resident(pk, M, K, t(shuf), t(qzeros), t(scales), 1, mode="out")
        np.testing.assert_allclose(y.float().cpu().numpy(), ref, rtol=2e-3, atol=2e-3 * np.abs(ref).max())
        assert torch.equal(y, slabs[0].to(torch.float16))
    # same K slices as the round-2 kernel: the slabs agree to fp32 rounding (bit for bit when the plan also splits K over
    # the waves the same way -- 4 waves x the same segments)
    if ops.wna16_ksplit(M, N, K, G) == ks:
        old, _ = ops.wna16_gemm_packed(pk, M, K, t(shuf), t(qzeros), t(scales), 1, partials=True)
        torch.testing.assert_close(slabs, old, rtol=2e-5, atol=2e-6 * float(np.abs(ref).max()))


@pytest.mark.parametrize("M", [1, 16, 32])
@pytest.mark.parametrize("K,N", SHAPES)
def test_resident_strip_layout_is_bit_identical(ops, K, N, M):
    shuf, qzeros, scales, a, _ = case(K, N)
    pk = ops.wna16_pack_a(t(a[:M]))
    qw = t(shuf)
    strip = ops.wna16_strip_relayout(qw, M, K // 128)
    assert strip.shape == qw.shape and not torch.equal(strip, qw)
    assert torch.equal(torch.sort(strip.flatten())[0], torch.sort(qw.flatten())[0])      # a permutation of the words
    s0, _ = ops.wna16_gemm_resident(pk, M, K, qw, t(qzeros), t(scales), 1, mode="slabs")
    s1, _ = ops.wna16_gemm_resident(pk, M, K, strip, t(qzeros), t(scales), 1, mode="slabs", strip_layout=True)
    assert torch.equal(s0, s1)
    # ... and the strip-major launch is the STREAM kernel for these plans (the one bench.py times): checked against the oracle
    # directly, not only through its bit-equality with the two-pass kernel
    ref = case(K, N)[4][:M]
    np.testing.assert_allclose(s1.double().sum(0).cpu().numpy(), ref, rtol=1e-4, atol=2e-5 * np.abs(ref).max())


@pytest.mark.parametrize("strip", [False, True])
@pytest.mark.parametrize("M", [1, 17, 32])
def test_resident_gate_up_silu_epilogue_vs_oracle(ops, M, strip):
    """gate_up with interleaved (gate_j, up_j) columns + SiluAndMul + pack in the epilogue, against
    silu_and_mul(round16(oracle GEMM)) (activation_kernels.cu:12-60 on the fp16-rounded projection)."""
    K, N = 4096, 28672
    shuf, qzeros, scales, a, ref = case(K, N)
    qw_i, qz_i, sc_i = ops.interleave_gate_up(t(shuf), t(qzeros), t(scales))
    pk = ops.wna16_pack_a(t(a[:M]))
    qw_k = ops.wna16_strip_relayout(qw_i, M, K // 128) if strip else qw_i
    act = ops.wna16_gemm_resident(pk, M, K, qw_k, qz_i, sc_i, 1, mode="silu", strip_layout=strip)
    got = unpack_a(act, M, N // 2).view(np.float16).astype(np.float64)
    gu = ref[:M].astype(np.float16)
    d = N // 2
    gate = gu[:, :d].astype(np.float64)
    silu = (gate / (1.0 + np.exp(-gate))).astype(np.float16).astype(np.float64)
    want = (silu * gu[:, d:].astype(np.float64)).astype(np.float16).astype(np.float64)
    np.testing.assert_allclose(got, want, rtol=4e-3, atol=4e-3 * np.abs(want).max())
    np.testing.assert_allclose(got, oa.silu_and_mul(ref[:M]), rtol=6e-3, atol=6e-3 * np.abs(want).max())
    # same partition as the round-2 kernel's fused form -> same bits
    # the round-2 kernel's fused form: same roundings, fp32 sums in a different order -> f16 results one ulp apart at most
    old = ops.wna16_gemm_silu_pack(pk, M, K, qw_i, qz_i, sc_i, 1)
    a_new = unpack_a(act, M, N // 2).view(np.float16).astype(np.float64)     # (rows >= M of a tile are not written)
    a_old = unpack_a(old, M, N // 2).view(np.float16).astype(np.float64)
    np.testing.assert_allclose(a_new, a_old, rtol=2e-3, atol=2e-3 * np.abs(want).max())


@pytest.mark.parametrize("cfg,K,N", [("4,4,0,3", 4096, 6144), ("4,8,1,0", 4096, 4096), ("4,2,1,0", 4096, 4096)])
def test_resident_alternative_configs_vs_oracle(ops, cfg, K, N):
    """The other instantiated (waves, segments per wave, 64-column passes, last-pass blocks) plans, picked by hand."""
    shuf, qzeros, scales, a, ref = case(K, N)
    os.environ["APHRO_WNA16_RES_CFG"] = cfg; ops.reload_env()
    for M in (5, 32):
        ks = ops.wna16_resident_ksplit(M, N, K, K // 128)
        assert ks >= 1
        pk = ops.wna16_pack_a(t(a[:M]))
        for strip in (False, True):
            qw = ops.wna16_strip_relayout(t(shuf), M, K // 128) if strip else t(shuf)
            slabs, _ = ops.wna16_gemm_resident(pk, M, K, qw, t(qzeros), t(scales), 1, mode="slabs", strip_layout=strip)
            s = slabs.double().sum(0).cpu().numpy()
            np.testing.assert_allclose(s, ref[:M], rtol=1e-4, atol=2e-5 * np.abs(ref).max())


@pytest.mark.parametrize("K,N", [(4096, 28672), (14336, 4096)])
def test_resident_bf16_vs_oracle(ops, K, N):
    """bf16 activations / scales / output: activations widened to f16 (saturating) when packed, scales read as bf16."""
    rng = np.random.default_rng(5)
    shuf, qzeros, _, _, _ = case(K, N)
    G = K // 128
    M = 32
    sc = torch.from_numpy((rng.uniform(0.75, 1.25, size=(G, N)) / (4.6 * np.sqrt(K))).astype(np.float32)).to(DEV).to(torch.bfloat16)
    a = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float32)).to(DEV).to(torch.bfloat16)
    a16 = a.float().cpu().numpy().astype(np.float16)          # what the packed buffer holds (|a| << 65504: exact widening? no: rounded)
    ref = oq.gptq_gemm(a.float().cpu().numpy(), shuf, qzeros, sc.float().cpu().numpy(), None, True)
    pk = ops.wna16_pack_a(a)
    slabs, ks = ops.wna16_gemm_resident(pk, M, K, t(shuf), t(qzeros), sc, 1, mode="slabs")
    s = slabs.double().sum(0).cpu().numpy()
    # bf16 -> f16 widening of the activations is exact for |a| in the f16 normal range (8 mantissa bits fit 11)
    np.testing.assert_allclose(s, ref, rtol=1e-4, atol=3e-5 * np.abs(ref).max())
    old, _ = ops.wna16_gemm_packed(pk, M, K, t(shuf), t(qzeros), sc, 1, partials=True)
    if old.shape == slabs.shape:
        torch.testing.assert_close(slabs, old, rtol=2e-5, atol=2e-6 * float(np.abs(ref).max()))


# ---- the op-level form: row-major activations, [M, N] out of ONE launch (aphro_wna16_gemm_rowmajor) ----------------------
@pytest.mark.parametrize("M", [1, 7, 16, 17, 32])
@pytest.mark.parametrize("K,N", SHAPES)
def test_rowmajor_one_launch_vs_oracle(ops, K, N, M):
    """What ``ops.gptq_gemm`` launches at M <= 32 for f16: the oracle bounds it; it is bit-identical to the slab form of the
    same kernel summed in slice order (the in-kernel reduce is arrival-order independent) and to itself over 20 runs (the
    tickets return to zero); a row pitch wider than K is read in place."""
    shuf, qzeros, scales, a, ref = case(K, N)
    ref = ref[:M]
    G = K // 128
    assert ops.wna16_gemm_rowmajor_supported(M, N, K, G, torch.float16)
    x = t(a[:M])
    y = ops.wna16_gemm_rowmajor(x, t(shuf), t(qzeros), t(scales), 1)
    np.testing.assert_allclose(y.float().cpu().numpy(), ref, rtol=2e-3, atol=2e-3 * np.abs(ref).max())
    slabs, ks = ops.wna16_gemm_resident(ops.wna16_pack_a(x), M, K, t(shuf), t(qzeros), t(scales), 1, mode="slabs")
    s = slabs[0].clone()
    for z in range(1, ks):
        s += slabs[z]
    assert torch.equal(y, s.to(torch.float16))
    for _ in range(20):
        assert torch.equal(ops.wna16_gemm_rowmajor(x, t(shuf), t(qzeros), t(scales), 1), y)
    wide = torch.zeros((M, K + 72), dtype=torch.float16, device=DEV)
    wide[:, 8:8 + K] = x
    assert torch.equal(ops.wna16_gemm_rowmajor(wide[:, 8:8 + K], t(shuf), t(qzeros), t(scales), 1), y)
    # the op itself takes this path ...
    g_idx = torch.empty(0, dtype=torch.int32, device=DEV)
    assert torch.equal(ops.gptq_gemm(x, t(shuf), t(qzeros), t(scales), g_idx, True, 4), y)
    # ... and agrees with the three-launch path it replaces (another K partition: fp32 rounding apart)
    os.environ["APHRO_WNA16_OP_NO_RESIDENT"] = "1"; ops.reload_env()
    try:
        old = ops.gptq_gemm(x, t(shuf), t(qzeros), t(scales), g_idx, True, 4)
    finally:
        os.environ.pop("APHRO_WNA16_OP_NO_RESIDENT"); ops.reload_env()
    torch.testing.assert_close(y.float(), old.float(), rtol=2e-3, atol=2e-3 * float(np.abs(ref).max()))


def test_rowmajor_one_launch_strip_layout_and_graph(ops):
    """Strip-major weights give the same bits; the launch is capturable (tickets allocated by the eager call before)."""
    K, N = 14336, 4096
    M = 32
    shuf, qzeros, scales, a, _ = case(K, N)
    x = t(a[:M])
    qw = t(shuf)
    y = ops.wna16_gemm_rowmajor(x, qw, t(qzeros), t(scales), 1)
    strip = ops.wna16_strip_relayout(qw, M, K // 128)
    assert torch.equal(ops.wna16_gemm_rowmajor(x, strip, t(qzeros), t(scales), 1, strip_layout=True), y)
    g_idx = torch.empty(0, dtype=torch.int32, device=DEV)
    qz, sc = t(qzeros), t(scales)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = ops.gptq_gemm(x, qw, qz, sc, g_idx, True, 4)
    for _ in range(5):
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, y)


def test_rowmajor_awq_and_bf16_routes(ops):
    """awq_gemm (zero offset 0) takes the same one-launch path; bf16 keeps the pack + GEMM path (the resident kernel
    computes on f16 fragments and has no in-register bf16 widening) -- both against the oracle."""
    rng = np.random.default_rng(5)
    K, N, M, G = 4096, 4096, 24, 32
    w = rng.integers(0, 16, size=(K, N), dtype=np.int32)
    zeros = rng.integers(0, 16, size=(G, N), dtype=np.int32)
    scales = (rng.random((G, N), dtype=np.float32) * 0.01 + 0.002).astype(np.float16)
    a = (rng.standard_normal((M, K)) * 0.5).astype(np.float16)
    qw = oq.awq_pack(w)
    qz = oq.awq_pack(zeros)
    ref = a.astype(np.float64) @ oq.awq_dequantize(qw, scales, qz).astype(np.float64)
    y = ops.awq_gemm(t(a), t(qw), t(scales), t(qz), 8)
    np.testing.assert_allclose(y.float().cpu().numpy(), ref, rtol=2e-3, atol=2e-3 * np.abs(ref).max())
    assert not ops.wna16_gemm_rowmajor_supported(M, N, K, G, torch.bfloat16)


# ---- 33..64 rows: the stream kernel on two 32-row halves (round 4; gridDim.z = 2, two workgroups per CU) -----------------
@pytest.mark.parametrize("M", [33, 48, 64])
@pytest.mark.parametrize("K,N", SHAPES + [(8192, 7168)])          # + the gate_up of one rank of Llama-3-70B at TP 8 (configs[3])
def test_row_halves_slabs_vs_oracle_and_vs_the_32_row_launches(ops, K, N, M):
    """M in 33..64 on strip-major weights: against the fp64 oracle (same bars as the 32-row test above), and bit for bit the
    32-row launch on rows 0..31 and the (M - 32)-row launch on the rest -- the halves ARE that kernel."""
    rng = np.random.default_rng(K + N + M)
    shuf, qzeros, scales, _, _ = case(K, N) if (K, N) in SHAPES else _case_shard(K, N)
    a = rng.standard_normal((M, K)).astype(np.float16)
    G = K // 128
    ks = ops.wna16_resident_ksplit(M, N, K, G)
    assert ks >= 1 and ks == ops.wna16_resident_ksplit(32, N, K, G)
    strip = ops.wna16_strip_relayout(t(shuf), M, G)
    assert torch.equal(strip, ops.wna16_strip_relayout(t(shuf), 32, G))         # one strip-major copy serves 1..64 rows
    pk = ops.wna16_pack_a(t(a))
    slabs, ks2 = ops.wna16_gemm_resident(pk, M, K, strip, t(qzeros), t(scales), 1, mode="slabs", strip_layout=True)
    assert ks2 == ks and slabs.shape == (ks, M, N)
    ref = oq.gptq_gemm(a, shuf, qzeros, scales, None, True)
    np.testing.assert_allclose(slabs.double().sum(0).cpu().numpy(), ref, rtol=1e-4, atol=2e-5 * np.abs(ref).max())
    lo, _ = ops.wna16_gemm_resident(ops.wna16_pack_a(t(a[:32])), 32, K, strip, t(qzeros), t(scales), 1, mode="slabs", strip_layout=True)
    hi, _ = ops.wna16_gemm_resident(ops.wna16_pack_a(t(a[32:])), M - 32, K, strip, t(qzeros), t(scales), 1, mode="slabs", strip_layout=True)
    assert torch.equal(slabs[:, :32], lo) and torch.equal(slabs[:, 32:], hi)


_SHARD_CASES = {}


def _case_shard(K, N):
    if (K, N) not in _SHARD_CASES:
        rng = np.random.default_rng(K * 7 + N)
        G = K // 128
        qweight = rng.integers(0, 2 ** 32, size=(K // 8, N), dtype=np.uint32).view(np.int32)
        qzeros = rng.integers(0, 2 ** 32, size=(G, N // 8), dtype=np.uint32).view(np.int32)
        scales = (rng.uniform(0.75, 1.25, size=(G, N)) / (4.6 * np.sqrt(K))).astype(np.float16)
        _SHARD_CASES[(K, N)] = (oq.gptq_shuffle(qweight), qzeros, scales, None, None)
    return _SHARD_CASES[(K, N)]


@pytest.mark.parametrize("M", [33, 64])
def test_row_halves_gate_up_silu_epilogue(ops, M):
    """gate_up + SiluAndMul + pack at 33..64 rows: the 32-row launches' packed rows bit for bit, the oracle at its bar."""
    K, N = 4096, 28672
    rng = np.random.default_rng(90 + M)
    shuf, qzeros, scales, _, _ = case(K, N)
    a = rng.standard_normal((M, K)).astype(np.float16)
    qw_i, qz_i, sc_i = ops.interleave_gate_up(t(shuf), t(qzeros), t(scales))
    strip = ops.wna16_strip_relayout(qw_i, M, K // 128)
    act = ops.wna16_gemm_resident(ops.wna16_pack_a(t(a)), M, K, strip, qz_i, sc_i, 1, mode="silu", strip_layout=True)
    got = unpack_a(act, M, N // 2)
    lo = unpack_a(ops.wna16_gemm_resident(ops.wna16_pack_a(t(a[:32])), 32, K, strip, qz_i, sc_i, 1, mode="silu", strip_layout=True), 32, N // 2)
    hi = unpack_a(ops.wna16_gemm_resident(ops.wna16_pack_a(t(a[32:])), M - 32, K, strip, qz_i, sc_i, 1, mode="silu", strip_layout=True),
                  M - 32, N // 2)
    assert np.array_equal(got[:32], lo) and np.array_equal(got[32:], hi)
    ref = oq.gptq_gemm(a, shuf, qzeros, scales, None, True)
    want = oa.silu_and_mul(ref)
    np.testing.assert_allclose(got.view(np.float16).astype(np.float64), want, rtol=6e-3, atol=6e-3 * np.abs(want).max())


def test_row_halves_decode_step_vs_the_mid_kernels(ops):
    """The fused decode step at batch 64 (3 layers of Llama-3-8B geometry): row halves on == off (the 33..64-row kernels) to the
    GEMMs' own rounding -- different K partitions, same arithmetic -- and the same greedy tokens."""
    from aphrodite_engine_amd import model as Mo
    from aphrodite_engine_amd.quantization.gptq import GPTQConfig
    cfg = Mo.LlamaConfig(hidden_size=4096, intermediate_size=14336, num_hidden_layers=3, num_attention_heads=32,
                         num_key_value_heads=8, vocab_size=1024, max_position_embeddings=2048)
    m = Mo.LlamaForCausalLM(cfg, GPTQConfig(4, 128, False), torch.float16, "auto").init_synthetic(DEV, seed=2)
    bs, ctx = 64, 90
    os.environ["APHRO_DECODE_ROW_HALVES"] = "1"          # (opt-in: the strip-major copies for 33..64 rows)
    try:
        for layer in m.layers:
            assert layer.enable_fused_silu(bs)
            assert layer.gate_up_strip is not None and {"qkv_proj", "o_proj", "down_proj"} <= set(layer.strip)
    finally:
        os.environ.pop("APHRO_DECODE_ROW_HALVES", None)
    meta, pos, nblocks = Mo.make_decode_metadata(bs, ctx, 16, DEV)
    kv0 = Mo.make_kv_caches(cfg, nblocks, 16, torch.float16, "auto", DEV)
    ids = torch.arange(bs, device=DEV) % cfg.vocab_size
    outs = []
    for on in ("1", "0"):
        os.environ["APHRO_DECODE_ROW_HALVES"] = on
        try:
            with torch.no_grad():
                outs.append(m(ids, pos, [c.clone() for c in kv0], meta).float())
        finally:
            os.environ.pop("APHRO_DECODE_ROW_HALVES", None)
    torch.testing.assert_close(outs[0], outs[1], rtol=2e-2, atol=2e-2 * float(outs[1].abs().max()))


def test_tp8_shard_layer_of_config3_takes_the_row_halves_and_slab_silu_path(ops):
    """ONE rank of Llama-3-70B at TP 8 (BASELINE configs[3]: AWQ shapes, batch 64), simulated on this GPU (all-reduces are
    identity: the rank's partial sums flow on): the fused decode layer -- K-sliced gate_up on the strip-major copy, two
    32-row halves, slabs into silu_and_mul_pack -- against the op-by-op path of the same rank at the tp2 test's bar."""
    import dataclasses
    from aphrodite_engine_amd import distributed as D
    from aphrodite_engine_amd import model as Mo
    from aphrodite_engine_amd.quantization.gptq import GPTQConfig
    D.init_simulated_tensor_parallel(8, 1.0)
    try:
        cfg = dataclasses.replace(Mo.LLAMA3_70B, num_hidden_layers=1, vocab_size=1024, max_position_embeddings=2048)
        with torch.no_grad():
            m = Mo.LlamaForCausalLM(cfg, GPTQConfig(4, 128, False), torch.float16, "auto").init_synthetic(DEV, seed=4)
            bs, ctx = 64, 70
            layer = m.layers[0]
            assert layer.tp == 8 and not layer.enable_fused_silu(bs)          # K-sliced gate_up: no SiluAndMul epilogue ...
            assert "gate_up_proj" in layer.strip                               # ... but a strip-major copy for the stream kernel
            assert ops.wna16_resident_ksplit(bs, layer.gate_up_proj.out_features, cfg.hidden_size, cfg.hidden_size // 128) == 4
            meta, pos, nblocks = Mo.make_decode_metadata(bs, ctx, 16, DEV)
            ids = torch.arange(bs, device=DEV) % cfg.vocab_size
            outs = []
            for fused in (False, True):
                kv = Mo.make_kv_caches(cfg, nblocks, 16, torch.float16, "auto", DEV, seed=3)
                m.use_fused_decode = fused
                if fused:
                    assert layer.fused_decode_ok(bs)
                outs.append(m(ids, pos, kv, meta).float())
        assert torch.isfinite(outs[1]).all()
        torch.testing.assert_close(outs[0], outs[1], atol=2e-2, rtol=2e-2)
    finally:
        D.destroy_tensor_parallel()
