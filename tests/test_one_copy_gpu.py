"""ONE resident copy of each int4 decode matrix (round 6): the strip-major order the <= 32-row decode
kernels stream is the only copy, and every other consumer reads IT -- the prompt-sized tile machines and the
dequantise-transpose pass address its 16-byte pieces in place.
Everything here is bit-for-bit against the same kernels on the [K/8, N] original (which the rest of the suite pins to the
oracle): a permutation of the weight words changes no arithmetic."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from aphrodite_engine_amd import _custom_ops
    return _custom_ops


def _weights(K, N, G, dtype, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // 8, N), generator=g, device=DEV, dtype=torch.int64).to(torch.int32)
    qz = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // G, N // 8), generator=g, device=DEV, dtype=torch.int64).to(torch.int32)
    sc = (torch.rand(K // G, N, generator=g, device=DEV) * 0.01 + 0.005).to(dtype)
    return qw, qz, sc, g


# the four Llama-3-8B shapes (strip geometries {4,8,1,3} gate_up, {4,7,1,0} down, {4,8,1,0}/{4,4,1,0} qkv / o: 64-column
# passes, 48-column remainder passes, one and several K slices) + a 70B TP-8 gate_up shard ({4,4,1,3}, four K slices)
SHAPES = [(4096, 28672), (14336, 4096), (4096, 6144), (4096, 4096), (8192, 7168)]


@pytest.mark.parametrize("K,N", SHAPES)
def test_strip_unrelayout_is_the_inverse_permutation(ops, K, N):
    qw, _, sc, _ = _weights(K, N, 128, torch.float16, K + N)
    geom = (torch.zeros(5, dtype=torch.int32)).numpy()
    from aphrodite_engine_amd import _lib
    assert _lib.lib().aphro_wna16_strip_geometry(32, N, K, K // 128, geom.ctypes.data) == 1
    nwv, nseg, np4, rem, ks = (int(x) for x in geom)
    assert N % (64 * np4 + 16 * rem) == 0 and (K // 128) % (nwv * nseg) == 0 and ks == (K // 128) // (nwv * nseg)
    st = ops.wna16_strip_relayout(qw, 32, K // 128)
    assert not torch.equal(st, qw)
    assert torch.equal(ops.wna16_strip_unrelayout(st, 32, K // 128), qw)
    # the 33..64-row class shares the layout (two 32-row halves of the same plan)
    if ops.wna16_resident_ksplit(64, N, K, K // 128) > 0:
        assert torch.equal(ops.wna16_strip_relayout(qw, 64, K // 128), st)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("K,N", SHAPES)
@pytest.mark.parametrize("M_", [300, 2048])
def test_large_gemm_reads_the_strip_major_copy_in_place(ops, dtype, K, N, M_):
    """The eight-phase fused form on strip-major weights == on [K/8, N], bit for bit (ragged last row tile, stream-K cuts)."""
    qw, qz, sc, g = _weights(K, N, 128, dtype, K + N + M_)
    a = (torch.randn(M_, K, generator=g, device=DEV) * 0.5).to(dtype)
    st = ops.wna16_strip_relayout(qw, 32, K // 128)
    want = ops._wna16_large(a, qw, qz, sc, None, 1)
    got = ops.wna16_gemm_large_strip(a, st, qz, sc, 1)
    assert torch.equal(got, want)
    # a strided activation view (lda > K) goes through unchanged
    wide = torch.zeros(M_, K + 64, dtype=dtype, device=DEV)
    wide[:, :K] = a
    assert torch.equal(ops.wna16_gemm_large_strip(wide[:, :K], st, qz, sc, 1), want)


@pytest.mark.parametrize("K,N", [(4096, 6144), (4096, 28672)])
def test_two_pass_form_dequantises_from_the_strip_major_copy(ops, K, N):
    """M = 8192: the dequantise-transpose pass reads the strip-major pieces; the fused form forced on the same call agrees."""
    qw, qz, sc, g = _weights(K, N, 128, torch.float16, K + N)
    a = (torch.randn(8192, K, generator=g, device=DEV) * 0.5).to(torch.float16)
    st = ops.wna16_strip_relayout(qw, 32, K // 128)
    want = ops._wna16_large(a, qw, qz, sc, None, 1)
    assert torch.equal(ops.wna16_gemm_large_strip(a, st, qz, sc, 1), want)
    with ops.knob("APHRO_WNA16_LARGE_TWO_PASS", 0):
        assert torch.equal(ops.wna16_gemm_large_strip(a, st, qz, sc, 1), want)


@pytest.mark.parametrize("M_,K,N", [(8192, 4096, 28672), (1000, 4096, 28672), (300, 4096, 14336)])
def test_silu_epilogue_on_the_strip_major_copy(ops, M_, K, N):
    qw, qz, sc, g = _weights(K, N, 128, torch.float16, K + N + 1)
    a = (torch.randn(M_, K, generator=g, device=DEV) * 0.5).to(torch.float16)
    if not ops.wna16_gemm_large_silu_supported(M_, N, K, K // 128) or ops.wna16_resident_ksplit(32, N, K, K // 128) <= 0:
        pytest.skip("shape is K-sliced by the plan / has no strip-major form")
    st = ops.wna16_strip_relayout(qw, 32, K // 128)
    assert torch.equal(ops.wna16_gemm_large_strip(a, st, qz, sc, 1, silu=True), ops.wna16_gemm_large_silu(a, qw, qz, sc, 1))


@pytest.mark.parametrize("M_,K,N", [(128, 4096, 4096), (256, 4096, 6144), (1024, 14336, 4096), (100, 1024, 14336), (513, 4096, 28672)])
def test_plans_outside_the_eight_phase_kernel_read_it_in_place_too(ops, M_, K, N):
    """Few-tile shapes run the one-workgroup-per-tile plans (K-sliced, 128- and 256-column tiles, two and three LDS stages) of
    the round-2 tile machine: its LDS-DMA weight loads take the strip-major addresses -- no extra workspace, same bits."""
    qw, qz, sc, g = _weights(K, N, 128, torch.float16, 5 + M_)
    a = (torch.randn(M_, K, generator=g, device=DEV) * 0.5).to(torch.float16)
    if ops.wna16_resident_ksplit(32, N, K, K // 128) <= 0:
        pytest.skip("no strip-major form")
    st = ops.wna16_strip_relayout(qw, 32, K // 128)
    from aphrodite_engine_amd import _lib
    lib = _lib.lib()
    assert lib.aphro_wna16_gemm_large_strip_workspace_bytes(M_, N, K, K // 128, 0, 32) == \
        lib.aphro_wna16_gemm_large_workspace_bytes(M_, N, K, K // 128, 0)
    assert torch.equal(ops.wna16_gemm_large_strip(a, st, qz, sc, 1), ops._wna16_large(a, qw, qz, sc, None, 1))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M_", [1, 17, 32, 33, 64, 100, 129, 700])
@pytest.mark.parametrize("K,N", [(4096, 6144), (14336, 4096)])
def test_linear_on_the_strip_major_copy_any_m(ops, M_, K, N, dtype):
    """ops.wna16_linear_strip (what a one-copy QuantLinear.forward runs) against gptq_gemm on the original: the same kernels
    where the same kernel serves both (bit-equal), the GEMMs' own rounding where the K partition differs.  bf16: no row-major
    one-launch form -- pack + the resident kernel; never the [K/8, N] rebuild on these shapes."""
    qw, qz, sc, g = _weights(K, N, 128, dtype, K + M_)
    a = (torch.randn(M_, K, generator=g, device=DEV) * 0.5).to(dtype)
    st = ops.wna16_strip_relayout(qw, 32, K // 128)
    calls = []
    real = ops.wna16_strip_unrelayout
    ops.wna16_strip_unrelayout = lambda *a_, **k_: (calls.append(1), real(*a_, **k_))[1]
    try:
        got = ops.wna16_linear_strip(a, st, qz, sc, 1)
    finally:
        ops.wna16_strip_unrelayout = real
    assert not calls
    want = ops.gptq_gemm(a, qw, qz, sc, torch.empty(0, dtype=torch.int32, device=DEV), True, 4)
    assert got.shape == want.shape and got.dtype == dtype
    tol = 2e-2 if dtype == torch.float16 else 6e-2
    torch.testing.assert_close(got.float(), want.float(), atol=tol, rtol=tol)
    if dtype == torch.float16 and (M_ > 128 or M_ <= 32):
        ref = ops._wna16_large(a, qw, qz, sc, None, 1) if M_ > 128 else ops.wna16_gemm_rowmajor(a, st, qz, sc, 1, strip_layout=True)
        assert torch.equal(got, ref)


@pytest.mark.parametrize("M_", [33, 48, 64])
@pytest.mark.parametrize("K,N", [(4096, 28672), (14336, 4096), (4096, 6144), (8192, 7168)])
def test_one_pass_kernel_on_the_strip_major_copy(ops, M_, K, N):
    """33..64 rows: wna16_gemm_mid_packed(strip_m=32) == the same launch on [K/8, N], bit for bit -- fp32 slabs, and the
    SiluAndMul + pack epilogue where the plan has one K slice (4 and 8 K-waves, K-sliced plans)."""
    qw, qz, sc, g = _weights(K, N, 128, torch.float16, K + N + M_)
    if ops.wna16_gemm_mid_ksplit(M_, N, K, K // 128) <= 0 or ops.wna16_resident_ksplit(32, N, K, K // 128) <= 0:
        pytest.skip("shape not served")
    a = (torch.randn(M_, K, generator=g, device=DEV) * 0.5).to(torch.float16)
    packed = ops.wna16_pack_a(a)
    st = ops.wna16_strip_relayout(qw, 32, K // 128)
    want, ks = ops.wna16_gemm_mid_packed(packed, M_, K, qw, qz, sc, 1, partials=True)
    got, ks2 = ops.wna16_gemm_mid_packed(packed, M_, K, st, qz, sc, 1, partials=True, strip_m=32)
    assert ks == ks2 and torch.equal(got, want)
    if ks == 1 and N % 256 == 0 and M_ % 16 == 0:      # (the packed output's padding rows are whatever torch.empty held)
        assert torch.equal(ops.wna16_gemm_mid_silu_pack(packed, M_, K, st, qz, sc, 1, strip_m=32),
                           ops.wna16_gemm_mid_silu_pack(packed, M_, K, qw, qz, sc, 1))


def _tiny_llama3(layers=2, seed=2):
    from aphrodite_engine_amd import model as Mo
    from aphrodite_engine_amd.quantization.gptq import GPTQConfig
    cfg = Mo.LlamaConfig(hidden_size=4096, intermediate_size=14336, num_hidden_layers=layers, num_attention_heads=32,
                         num_key_value_heads=8, vocab_size=1024, max_position_embeddings=2048)
    return Mo, cfg, Mo.LlamaForCausalLM(cfg, GPTQConfig(4, 128, False), torch.float16, "auto").init_synthetic(DEV, seed=seed)


def test_model_with_one_copy_is_the_model_with_two(ops):
    """Two layers of Llama-3-8B geometry: enable_one_copy releases the [K/8, N] words (the allocator sees it), and the decode
    step at 32 rows and a 300-token prompt give the SAME logits as before; 48-row decode and a 40-token prompt (32-row passes
    of the stream kernel instead of the one-pass kernel) the same greedy tokens; restore_op_level_layouts brings the checkpoint's words back bit for bit."""
    Mo, cfg, m = _tiny_llama3()
    with torch.no_grad():
        orig = {(i, n): getattr(l, n).qweight.data.clone() for i, l in enumerate(m.layers)
                for n in ("qkv_proj", "o_proj", "gate_up_proj", "down_proj")}
        for layer in m.layers:
            assert layer.enable_fused_silu(32, keep_original=False)

        def decode(bs, ctx=90):
            meta, pos, nblocks = Mo.make_decode_metadata(bs, ctx, 16, DEV)
            kv = Mo.make_kv_caches(cfg, nblocks, 16, torch.float16, "auto", DEV, seed=3)
            ids = torch.arange(bs, device=DEV) % cfg.vocab_size
            assert all(l.fused_decode_ok(bs) for l in m.layers)
            return m(ids, pos, kv, meta).float()

        def prefill(T):
            from aphrodite_engine_amd.attention.backend import MI355XAttentionMetadata
            nblk = (T + 15) // 16
            kv = Mo.make_kv_caches(cfg, nblk, 16, torch.float16, "auto", DEV, fill=False)
            bt = torch.arange(nblk, device=DEV, dtype=torch.int32).flip(0).view(1, nblk)
            pos = torch.arange(T, device=DEV, dtype=torch.int64)
            slots = bt[0, pos // 16].long() * 16 + pos % 16
            i32 = lambda *a: torch.tensor(a, dtype=torch.int32, device=DEV)
            meta = MI355XAttentionMetadata(
                num_prefills=1, num_prefill_tokens=T, num_decode_tokens=0, slot_mapping=slots, seq_lens=[T],
                seq_lens_tensor=i32(T), max_query_len=T, max_prefill_seq_len=T, max_decode_seq_len=0,
                query_start_loc=i32(0, T), seq_start_loc=i32(0, T), context_lens_tensor=i32(0), block_tables=bt,
                use_cuda_graph=False, max_context_len=0)
            ids = (torch.arange(T, device=DEV) * 7) % cfg.vocab_size
            return m(ids, pos, kv, meta).float()

        before = dict(d32=decode(32), d48=decode(48), p300=prefill(300), p40=prefill(40))
        torch.cuda.synchronize()
        mem0 = torch.cuda.memory_allocated()
        freed = sum(layer.enable_one_copy() for layer in m.layers)
        torch.cuda.synchronize()
        per_layer = (4096 * 6144 + 4096 * 4096 + 4096 * 28672 + 14336 * 4096) // 2
        assert freed == len(m.layers) * per_layer
        assert mem0 - torch.cuda.memory_allocated() == freed
        assert all(l.one_copy and l.enable_one_copy() == 0 for l in m.layers)
        after = dict(d32=decode(32), d48=decode(48), p300=prefill(300), p40=prefill(40))
        for k in ("d32", "p300"):                   # same kernels, same K partitions: same bits
            assert torch.equal(before[k], after[k]), k
        for k in ("d48", "p40"):                    # 33..64 rows: 32-row passes of the stream kernel instead of the one-pass
            torch.testing.assert_close(after[k], before[k], atol=3e-2, rtol=3e-2)       # kernel -- other K partitions
            assert torch.equal(after[k].argmax(-1), before[k].argmax(-1)), k
        for layer in m.layers:
            layer.restore_op_level_layouts()
            assert not layer.one_copy
        for (i, n), w in orig.items():
            assert torch.equal(getattr(m.layers[i], n).qweight.data, w), (i, n)


def test_one_copy_is_refused_where_the_step_reads_the_row_order(ops):
    """TP shards and sparse layers run round-2 kernels on [K/8, N]: enable_one_copy leaves them alone."""
    Mo, cfg, m = _tiny_llama3(layers=1)
    layer = m.layers[0]
    os.environ["APHRO_WEIGHTS_TWO_COPIES"] = "1"
    try:
        assert layer.enable_fused_silu(32, keep_original=False) and layer.enable_one_copy() == 0 and not layer.one_copy
    finally:
        os.environ.pop("APHRO_WEIGHTS_TWO_COPIES")
    assert layer.enable_one_copy() > 0
    assert layer.enable_fused_silu(32, keep_original=False)         # (rebuilding the layouts undoes it first)
    assert not layer.one_copy and not getattr(layer.qkv_proj, "qweight_strip_major", False)


def test_tp_shard_keeps_no_copy_its_step_never_reads(ops):
    """ONE rank of Llama-3-70B at TP 8, layouts for 32 rows: the row-parallel projections (o_proj, down_proj) run the round-2
    kernel on [K/8, N] under TP -- no strip-major copy is built for them (14.7 MB per layer that used to sit unread); the
    column-parallel ones keep theirs; the fused decode step still agrees with the op-by-op path."""
    import dataclasses
    from aphrodite_engine_amd import distributed as D
    from aphrodite_engine_amd import model as Mo
    from aphrodite_engine_amd.quantization.gptq import GPTQConfig
    D.init_simulated_tensor_parallel(8, 1.0)
    try:
        cfg = dataclasses.replace(Mo.LLAMA3_70B, num_hidden_layers=1, vocab_size=1024, max_position_embeddings=2048)
        with torch.no_grad():
            m = Mo.LlamaForCausalLM(cfg, GPTQConfig(4, 128, False), torch.float16, "auto").init_synthetic(DEV, seed=4)
            bs, ctx = 32, 70
            layer = m.layers[0]
            layer.enable_fused_silu(bs)
            assert layer.tp == 8 and not ({"o_proj", "down_proj"} & set(layer.strip))
            assert layer.enable_one_copy() == 0 and not layer.one_copy            # (TP: [K/8, N] stays)
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


@pytest.mark.parametrize("K,N", [(4096, 28672), (14336, 4096)])
def test_awq_prepacked_words_zero_offset_0(ops, K, N):
    """The AWQ-prepacked form of the same layout (zero points without GPTQ's + 1): every strip-major entry against its
    [K/8, N] twin, bit for bit -- prompt-sized, 33..64 rows, <= 32 rows."""
    qw, qz, sc, g = _weights(K, N, 128, torch.float16, K + 3)
    st = ops.wna16_strip_relayout(qw, 32, K // 128)
    a = (torch.randn(700, K, generator=g, device=DEV) * 0.5).to(torch.float16)
    assert torch.equal(ops.wna16_gemm_large_strip(a, st, qz, sc, 0), ops._wna16_large(a, qw, qz, sc, None, 0))
    assert not torch.equal(ops.wna16_gemm_large_strip(a, st, qz, sc, 0), ops.wna16_gemm_large_strip(a, st, qz, sc, 1))
    p48 = ops.wna16_pack_a(a[:48])
    want, _ = ops.wna16_gemm_mid_packed(p48, 48, K, qw, qz, sc, 0, partials=True)
    got, _ = ops.wna16_gemm_mid_packed(p48, 48, K, st, qz, sc, 0, partials=True, strip_m=32)
    assert torch.equal(got, want)
    assert torch.equal(ops.wna16_linear_strip(a[:17], st, qz, sc, 0), ops.wna16_gemm_rowmajor(a[:17], qw, qz, sc, 0))
