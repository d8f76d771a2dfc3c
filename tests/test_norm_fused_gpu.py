"""Norm-in-consumer launch (round 4, csrc/wna16_gemm_resident.hip ``wna16_gemm_stream_kernel<.., NORM_T, STAGE>``,
``aphro_wna16_gemm_norm_fused``; opt-in on the decode path, see profiles/r4_norm_in_consumer.txt): split-K slab reduce +
fused_add_rms_norm (kernels/layernorm_kernels.cu:200-240) + pack produced by the first M workgroups of the W4A16 GEMM
launch that consumes them (q_gemm.cu:190-326 role), at the configs[1] gate_up shape (4096 x 28672 + SiluAndMul).

Checked against the ORACLE (oracle.attention.fused_add_rms_norm -> oracle.quant.gptq_gemm) at the reference's bars, and bit
for bit against the two separate launches (each oracle-checked on its own in tests/test_ops_gpu.py /
tests/test_resident_gpu.py) -- including many back-to-back launches on changing data and HIP-graph replays, which is where a
stale cache line or a missed hand-over would show."""
import numpy as np
import pytest
import torch

from oracle import attention as oa
from oracle import quant as oq
from tests.test_headline_gpu import case, t, unpack_a

pytestmark = pytest.mark.gpu
DEV = "cuda"
EPS = 1e-5


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X (torch.cuda.is_available() is False)")
    from aphrodite_engine_amd import _custom_ops, _lib
    _lib.lib()
    return _custom_ops


def _inputs(M, K, seed, dtype=torch.float16):
    g = torch.Generator(device=DEV)
    g.manual_seed(seed)
    slabs = torch.randn((4, M, K), generator=g, device=DEV, dtype=torch.float32) * 0.5
    residual = (torch.randn((M, K), generator=g, device=DEV, dtype=torch.float32)).to(dtype)
    weight = (1.0 + 0.1 * torch.randn((K,), generator=g, device=DEV, dtype=torch.float32)).to(dtype)
    return slabs, residual, weight


def _two_launches(ops, slabs, residual, weight, M, K, strip, qz, sc, mode):
    res = residual.clone()
    packed, _ = ops.fused_add_rms_norm_pack(None, slabs, res, True, weight, EPS)
    out = ops.wna16_gemm_resident(packed, M, K, strip, qz, sc, 1, mode=mode, strip_layout=True)
    return (out if mode == "silu" else out[0]), res, packed


def _weights(ops, name, M):
    K, N = (4096, 6144) if name == "qkv" else (4096, 28672)
    shuf, qzeros, scales, _, _ = case(K, N)
    qw, qz, sc = t(shuf), t(qzeros), t(scales)
    if name == "gate_up":
        qw, qz, sc = ops.interleave_gate_up(qw, qz, sc)
    strip = ops.wna16_strip_relayout(qw, M, K // 128)
    return K, N, strip, qz, sc, qw


@pytest.mark.parametrize("M", [1, 5, 16, 17, 32])
@pytest.mark.parametrize("name", ["gate_up"])
def test_norm_fused_is_the_two_launches_bit_for_bit(ops, name, M):
    K, N, strip, qz, sc, _ = _weights(ops, name, M)
    mode = "silu" if name == "gate_up" else "slabs"
    assert ops.wna16_gemm_norm_fused_supported(M, N, K, K // 128, 4, torch.float16)
    sync = torch.zeros(1, dtype=torch.int32, device=DEV)
    for rep in range(6):        # fresh data every time, the same buffers underneath (the caching allocator reuses them)
        slabs, residual, weight = _inputs(M, K, 100 * M + rep)
        want, want_res, _ = _two_launches(ops, slabs, residual, weight, M, K, strip, qz, sc, mode)
        res = residual.clone()
        sync.zero_()
        got = ops.wna16_gemm_norm_fused(slabs, res, weight, EPS, strip, qz, sc, 1, sync, mode=mode)
        got = got if mode == "silu" else got[0]
        torch.cuda.synchronize()
        assert int(sync.item()) == M                       # every producer arrived exactly once
        assert torch.equal(res, want_res)
        if mode == "silu":
            assert np.array_equal(unpack_a(got, M, N // 2), unpack_a(want, M, N // 2))
        else:
            assert torch.equal(got, want)


@pytest.mark.parametrize("name", ["gate_up"])
def test_norm_fused_vs_oracle(ops, name):
    """oracle fused_add_rms_norm -> oracle GEMM (-> SiluAndMul) on the same inputs."""
    M = 32
    K, N, strip, qz, sc, qw = _weights(ops, name, M)
    slabs, residual, weight = _inputs(M, K, 7)
    x_np = (slabs[0] + slabs[1] + slabs[2] + slabs[3]).to(torch.float16).cpu().numpy()      # (slab order, as the kernel adds them)
    y_ref, res_ref = oa.fused_add_rms_norm(x_np, residual.cpu().numpy(), weight.cpu().numpy(), EPS)
    res = residual.clone()
    sync = torch.zeros(1, dtype=torch.int32, device=DEV)
    mode = "silu" if name == "gate_up" else "slabs"
    got = ops.wna16_gemm_norm_fused(slabs, res, weight, EPS, strip, qz, sc, 1, sync, mode=mode)
    np.testing.assert_allclose(res.float().cpu().numpy(), np.asarray(res_ref, np.float32), rtol=1e-3, atol=1e-3)
    shuf, qzeros, scales, _, _ = case(K, N)
    ref = oq.gptq_gemm(np.asarray(y_ref, np.float16), shuf, qzeros, scales, None, True)      # fp64 [M, N], [gate | up] columns
    if mode == "slabs":
        s = got[0].double().sum(0).cpu().numpy()
        np.testing.assert_allclose(s, ref, rtol=2e-3, atol=2e-3 * np.abs(ref).max())        # one f16 ulp of y moves the sum
    else:
        want = oa.silu_and_mul(ref)
        g = unpack_a(got, M, N // 2).view(np.float16).astype(np.float64)
        np.testing.assert_allclose(g, want, rtol=8e-3, atol=8e-3 * np.abs(want).max())


def test_norm_fused_bf16_is_the_two_launches_bit_for_bit(ops):
    M, K, N = 32, 4096, 28672
    rng = np.random.default_rng(5)
    shuf, qzeros, _, _, _ = case(K, N)
    sc = torch.from_numpy((rng.uniform(0.75, 1.25, size=(K // 128, N)) / (4.6 * np.sqrt(K))).astype(np.float32)).to(DEV).to(torch.bfloat16)
    qw, qz, sc = ops.interleave_gate_up(t(shuf), t(qzeros), sc)
    strip = ops.wna16_strip_relayout(qw, M, K // 128)
    slabs, residual, weight = _inputs(M, K, 11, torch.bfloat16)
    want, want_res, _ = _two_launches(ops, slabs, residual, weight, M, K, strip, qz, sc, "silu")
    res = residual.clone()
    sync = torch.zeros(1, dtype=torch.int32, device=DEV)
    got = ops.wna16_gemm_norm_fused(slabs, res, weight, EPS, strip, qz, sc, 1, sync, mode="silu")
    assert torch.equal(res, want_res)
    assert np.array_equal(unpack_a(got, M, N // 2), unpack_a(want, M, N // 2))


def test_norm_fused_refuses_what_it_does_not_serve(ops):
    assert not ops.wna16_gemm_norm_fused_supported(32, 6144, 4096, 32, 4, torch.float16)       # qkv: no instantiation
    assert not ops.wna16_gemm_norm_fused_supported(33, 28672, 4096, 32, 4, torch.float16)      # > 32 rows
    assert not ops.wna16_gemm_norm_fused_supported(32, 28672, 4096, 32, 2, torch.float16)      # two input slabs
    assert not ops.wna16_gemm_norm_fused_supported(32, 28672, 8192, 64, 4, torch.float16)      # hidden != 4096


def test_norm_fused_chain_under_graph_replay(ops):
    """A chain of norm + gate_up launches, each with its own ticket, ONE fill zeroing all tickets at the top (the decode
    step's arrangement), captured once and replayed with new inputs: every replay must reproduce the eager two-launch
    results of ITS inputs bit for bit (buffers are reused across replays: a stale line or a consumer running ahead of its
    producers would show up as the previous replay's values)."""
    M = 32
    K, Ng, strip_g, qz_g, sc_g, _ = _weights(ops, "gate_up", M)
    slabs = torch.zeros((4, M, K), dtype=torch.float32, device=DEV)
    residual = torch.zeros((M, K), dtype=torch.float16, device=DEV)
    _, _, weight = _inputs(M, K, 3)
    sync = torch.zeros(6, dtype=torch.int32, device=DEV)

    def chain():
        sync.zero_()
        outs = []
        res = residual.clone()
        for i in range(6):
            outs.append(ops.wna16_gemm_norm_fused(slabs, res, weight, EPS, strip_g, qz_g, sc_g, 1, sync[i:i + 1], mode="silu"))
        return outs, res

    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        chain()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            outs, res_out = chain()
    for rep in range(5):
        sl, rs, _ = _inputs(M, K, 1000 + rep)
        slabs.copy_(sl)
        residual.copy_(rs)
        graph.replay()
        torch.cuda.synchronize()
        res = residual.clone()
        for i in range(6):
            packed, _ = ops.fused_add_rms_norm_pack(None, slabs, res, True, weight, EPS)
            a = ops.wna16_gemm_resident(packed, M, K, strip_g, qz_g, sc_g, 1, mode="silu", strip_layout=True)
            assert np.array_equal(unpack_a(outs[i], M, Ng // 2), unpack_a(a, M, Ng // 2)), f"replay {rep} link {i}: gate_up differs"
        assert torch.equal(res_out, res)
        assert sync.tolist() == [M] * 6


def test_fused_decode_step_with_and_without_norm_fused(ops):
    """The whole fused decode step of a 3-layer Llama-3-8B-geometry model: norm-in-consumer on == off, token for token and
    hidden state for hidden state (bit for bit)."""
    import os
    from aphrodite_engine_amd import model as Mo
    from aphrodite_engine_amd.quantization.gptq import GPTQConfig
    cfg = Mo.LlamaConfig(hidden_size=4096, intermediate_size=14336, num_hidden_layers=3, num_attention_heads=32,
                         num_key_value_heads=8, vocab_size=1024, max_position_embeddings=2048)
    m = Mo.LlamaForCausalLM(cfg, GPTQConfig(4, 128, False), torch.float16, "auto").init_synthetic(DEV, seed=1)
    bs, ctx = 32, 100
    for layer in m.layers:
        assert layer.enable_fused_silu(bs)
    meta, pos, nblocks = Mo.make_decode_metadata(bs, ctx, 16, DEV)
    kv0 = Mo.make_kv_caches(cfg, nblocks, 16, torch.float16, "auto", DEV)
    ids = torch.arange(bs, device=DEV) % cfg.vocab_size
    outs = []
    for on in (False, True):
        m.norm_fused = on
        kv = [c.clone() for c in kv0]
        with torch.no_grad():
            outs.append(m(ids, pos, kv, meta).clone())
    if m._norm_sync is None:
        pytest.fail("the norm-in-consumer path did not run")
    assert m._norm_sync.tolist() == [bs] * 3         # every layer's post-attention norm rode in its gate_up launch
    assert torch.equal(outs[0], outs[1])
