"""On-disk format ingestion, GPU half: a synthetic checkpoint of every supported format is loaded
onto the MI355X (post-load repack / requantisation included) and each linear of the loaded model is
checked against x @ W with W dequantised from the LOGICAL matrices the writer kept; then a decode
step runs through the loaded model (fused fast path where the format has one)."""
import numpy as np
import pytest
import torch

from aphrodite_engine_amd import loader as L
from aphrodite_engine_amd import model as M
from tests import ckpt_util as CU

pytestmark = pytest.mark.gpu
DEV = "cuda"
CFG = M.TINY
PROJ_OF = {"qkv_proj": ["self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj"], "o_proj": ["self_attn.o_proj"],
           "gate_up_proj": ["mlp.gate_proj", "mlp.up_proj"], "down_proj": ["mlp.down_proj"]}


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from aphrodite_engine_amd import _custom_ops
    return _custom_ops


def dense_weight(fmt, lg):
    """Logical matrices -> float32 [K, N]."""
    if fmt == "fp16":
        return lg["w"].astype(np.float32)
    if fmt in ("gptq", "awq"):
        g = np.arange(lg["q"].shape[0]) // 128
        return (lg["q"] - lg["zp"][g]).astype(np.float32) * lg["s"].astype(np.float32)[g]
    if fmt in ("ct-w4a16", "ct-w8a16i"):
        g = np.arange(lg["q"].shape[0]) // 128
        return (lg["q"] - (8 if fmt == "ct-w4a16" else 128)).astype(np.float32) * lg["s"].astype(np.float32)[g]
    w = lg["wq"].float() * lg["s"].float().reshape(-1, 1) if lg["s"].numel() > 1 else lg["wq"].float() * lg["s"]
    return w.numpy().T      # stored [N, K]


@pytest.mark.parametrize("fmt", ["fp16", "gptq", "awq", "fp8", "ct-fp8-channel", "ct-fp8-tensor", "ct-w8a16",
                                 "ct-w4a16", "ct-w8a16i"])
def test_loaded_checkpoint_linears_and_decode(ops, tmp_path, fmt):
    truth = CU.write_checkpoint(str(tmp_path), CFG, fmt, seed=11,
                                kv_scales="kv" if fmt == "fp8" else None)
    kv_dtype = "fp8" if fmt == "fp8" else "auto"
    with torch.no_grad():
        m = L.load_model(str(tmp_path), dtype=torch.float16, kv_cache_dtype=kv_dtype, device=DEV)
        rng = np.random.default_rng(5)
        for li, layer in enumerate(m.layers):
            for mod, projs in PROJ_OF.items():
                lin = getattr(layer, mod)
                w = np.concatenate([dense_weight(fmt, truth["logical"][f"model.layers.{li}.{p}"]) for p in projs], 1)
                x = torch.from_numpy(rng.standard_normal((7, w.shape[0])).astype(np.float32)).half().to(DEV)
                got = lin(x).float().cpu().numpy()
                ref = x.float().cpu().numpy() @ w
                err = np.abs(got - ref).mean() / np.abs(ref).mean()
                # W8A8 adds the activation quantisation (e4m3: 2^-4 relative steps) on top of fp16 rounding
                assert err < (0.04 if fmt in ("fp8", "ct-fp8-channel", "ct-fp8-tensor") else 4e-3), (fmt, mod, err)
        if fmt == "fp8":
            assert m.layers[1].k_scale == pytest.approx(0.021) and m.layers[1].v_scale == pytest.approx(0.031)
        # one decode step through the loaded model: fused fast path == op-by-op path
        meta, pos, nblocks = M.make_decode_metadata(5, [3, 17, 64, 200, 129], 16, DEV)
        ids = torch.randint(0, CFG.vocab_size, (5, ), device=DEV)
        outs = []
        for fused in (False, True):
            caches = M.make_kv_caches(CFG, nblocks, 16, torch.float16, kv_dtype, DEV, seed=3)
            m.use_fused_decode = fused
            outs.append(m(ids, pos, caches, meta).float())
        assert torch.isfinite(outs[0]).all()
        if fmt in ("gptq", "awq", "ct-w4a16"):
            assert all(l.fused_decode_ok(5) for l in m.layers)
            torch.testing.assert_close(outs[0], outs[1], atol=2e-2, rtol=2e-2)
        elif fmt in ("ct-fp8-channel", "ct-fp8-tensor"):
            assert all(l.fused_decode_fp8_ok(5) for l in m.layers)
            assert torch.equal(outs[0], outs[1])
        else:
            assert torch.equal(outs[0], outs[1])      # no fast path for this format: same code twice
        logits = m.compute_logits(outs[1].half())
        assert logits.shape == (5, CFG.vocab_size) and torch.isfinite(logits.float()).all()


@pytest.mark.parametrize("fmt", ["fp16", "gptq", "ct-fp8-channel"])
def test_loaded_checkpoint_with_projection_biases(ops, tmp_path, fmt):
    """config.attention_bias / mlp_bias (models/llama.py:62-82, 135-150): every projection = x W + b through the quant method's
    apply(layer, x, bias); the layers stay off the fused steps (which have no place for a bias), a decode step runs op by op
    and differs from the same checkpoint with its biases zeroed."""
    truth = CU.write_checkpoint(str(tmp_path), CFG, fmt, seed=12, proj_bias="both")
    with torch.no_grad():
        m = L.load_model(str(tmp_path), dtype=torch.float16, device=DEV)
        rng = np.random.default_rng(6)
        for li, layer in enumerate(m.layers):
            assert layer.has_bias and not layer.fused_decode_ok(5) and not layer.fused_decode_fp8_ok(5)
            for mod, projs in PROJ_OF.items():
                lin = getattr(layer, mod)
                w = np.concatenate([dense_weight(fmt, truth["logical"][f"model.layers.{li}.{p}"]) for p in projs], 1)
                b = np.concatenate([truth["logical"][f"model.layers.{li}.{p}"]["bias"].astype(np.float32) for p in projs])
                x = torch.from_numpy(rng.standard_normal((7, w.shape[0])).astype(np.float32)).half().to(DEV)
                got = lin(x).float().cpu().numpy()
                ref = x.float().cpu().numpy() @ w + b
                err = np.abs(got - ref).mean() / np.abs(ref).mean()
                assert err < (0.04 if fmt == "ct-fp8-channel" else 4e-3), (fmt, mod, err)
                nob = lin(x, add_bias=False).float().cpu().numpy()
                np.testing.assert_allclose(got - nob, np.broadcast_to(b, got.shape), atol=2e-2)
        meta, pos, nblocks = M.make_decode_metadata(5, [3, 17, 64, 200, 129], 16, DEV)
        ids = torch.randint(0, CFG.vocab_size, (5, ), device=DEV)

        def step():
            caches = M.make_kv_caches(CFG, nblocks, 16, torch.float16, "auto", DEV, seed=3)
            return m(ids, pos, caches, meta).float()
        with_bias = step()
        assert torch.isfinite(with_bias).all()
        for layer in m.layers:
            for lin in layer.linears():
                lin.bias.data.zero_()
        assert not torch.equal(with_bias, step())


def test_loaded_gptq_matches_directly_built_model(ops, tmp_path):
    """The loader path and direct parameter assignment give the same model, bit for bit."""
    from aphrodite_engine_amd.quantization.gptq import GPTQConfig
    truth = CU.write_checkpoint(str(tmp_path), CFG, "gptq", seed=12)
    t = truth["tensors"]
    with torch.no_grad():
        a = L.load_model(str(tmp_path), device=DEV)
        b = M.LlamaForCausalLM(CFG, GPTQConfig(4, 128, False), torch.float16)
        b.embed_tokens.copy_(t["model.embed_tokens.weight"])
        b.norm.copy_(t["model.norm.weight"])
        b.lm_head.copy_(t["lm_head.weight"])
        for li, layer in enumerate(b.layers):
            base = f"model.layers.{li}."
            layer.input_layernorm.copy_(t[base + "input_layernorm.weight"])
            layer.post_attention_layernorm.copy_(t[base + "post_attention_layernorm.weight"])
            for mod, projs in PROJ_OF.items():
                lin = getattr(layer, mod)
                for attr, dim in (("qweight", 1), ("qzeros", 1), ("scales", 1)):
                    getattr(lin, attr).copy_(torch.cat([t[base + p + "." + attr] for p in projs], dim))
                lin.g_idx.copy_(t[base + projs[0] + ".g_idx"])
        b.to(DEV)
        b.cos_sin = a.cos_sin
        b.process_weights_after_loading()
        meta, pos, nblocks = M.make_decode_metadata(4, [5, 33, 100, 64], 16, DEV)
        ids = torch.randint(0, CFG.vocab_size, (4, ), device=DEV)
        outs = []
        for m in (a, b):
            caches = M.make_kv_caches(CFG, nblocks, 16, torch.float16, "auto", DEV, seed=3)
            outs.append(m(ids, pos, caches, meta))
        assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("fmt", ["gptq", "awq", "ct-w4a16"])
def test_loaded_mixtral_experts_match_dense_reference(ops, tmp_path, fmt):
    """Mixtral-style int4 checkpoint (GPTQ, AWQ, or compressed-tensors pack-quantized through CompressedTensorsMoEMethod)
    -> FusedMoE over the grouped CDNA4 GEMM.  The sparse block of
    every layer is compared with a dense fp32 restatement of MixtralMoE (softmax -> top-k ->
    renormalise -> per-expert SiluAndMul MLP, modeling/models/mixtral_quant.py:91-156) on the weights
    dequantised from the writer's logical matrices; then decode runs fused == op-by-op."""
    cfg = M.TINY_MOE
    truth = CU.write_checkpoint(str(tmp_path), cfg, fmt, seed=31)
    lg, t = truth["logical"], truth["tensors"]
    with torch.no_grad():
        m = L.load_model(str(tmp_path), device=DEV)
        rng = np.random.default_rng(6)
        for li, layer in enumerate(m.layers):
            assert not hasattr(layer.experts, "w13_qweight") and not hasattr(layer.experts, "w13_weight_packed")   # checkpoint layout dropped after the repack
            x = torch.from_numpy(rng.standard_normal((9, cfg.hidden_size)).astype(np.float32)).half().to(DEV)
            got = layer.moe_block(x).float().cpu()
            xf = x.float().cpu()
            gate = t[f"model.layers.{li}.block_sparse_moe.gate.weight"].float()
            logits = (x @ gate.half().to(DEV).t()).float().cpu()         # the router GEMM is fp16 in the model too
            probs = torch.softmax(logits, dim=-1)
            w, ids = torch.topk(probs, cfg.num_experts_per_tok, dim=-1)
            w = w / w.sum(dim=-1, keepdim=True)
            ref = torch.zeros_like(xf)
            for e in range(cfg.num_local_experts):
                base = f"model.layers.{li}.block_sparse_moe.experts.{e}."
                w1, w3, w2 = (torch.from_numpy(dense_weight(fmt, lg[base + n])) for n in ("w1", "w3", "w2"))
                h = torch.nn.functional.silu(xf @ w1) * (xf @ w3)
                y = h @ w2
                ref += y * (w * (ids == e)).sum(dim=-1, keepdim=True)
            err = (got - ref).abs().mean() / ref.abs().mean()
            assert err < 1e-2, (fmt, li, float(err))
        meta, pos, nblocks = M.make_decode_metadata(5, [3, 17, 64, 200, 129], 16, DEV)
        ids_ = torch.randint(0, cfg.vocab_size, (5, ), device=DEV)
        outs = []
        for fused in (False, True):
            caches = M.make_kv_caches(cfg, nblocks, 16, torch.float16, "auto", DEV, seed=3)
            m.use_fused_decode = fused
            if fmt != "ct-w4a16":      # (the compressed-tensors dense projections run through the MPLinearKernel seam)
                assert all(l.fused_decode_ok(5) for l in m.layers)
            outs.append(m(ids_, pos, caches, meta).float())
        assert torch.isfinite(outs[0]).all()
        torch.testing.assert_close(outs[0], outs[1], atol=2e-2, rtol=2e-2)


def test_synthetic_mixtral_decode_runs(ops):
    from aphrodite_engine_amd.quantization.gptq import GPTQConfig
    with torch.no_grad():
        m = M.LlamaForCausalLM(M.TINY_MOE, GPTQConfig(4, 128, False), torch.float16).init_synthetic(torch.device(DEV))
        meta, pos, nblocks = M.make_decode_metadata(32, 300, 16, DEV)
        caches = M.make_kv_caches(M.TINY_MOE, nblocks, 16, torch.float16, "auto", DEV)
        ids = torch.randint(0, M.TINY_MOE.vocab_size, (32, ), device=DEV)
        out = m(ids, pos, caches, meta)
        assert out.shape == (32, M.TINY_MOE.hidden_size) and torch.isfinite(out.float()).all()
        assert m.weight_bytes_per_layer() > 0


def test_reference_model_adapter_matches_load_model(ops, tmp_path):
    """reference_model.MI355XLlamaForCausalLM -- the class the plugin hands to the reference's ModelRegistry -- driven the
    way the reference drives a model: constructed from a HF-style config + cache config + quant config
    (model_loader/loader.py:144-157), ``load_weights`` with the checkpoint's (name, tensor) pairs, the loader's post-load
    pass over modules with a ``quant_method`` (:402-408), then ``forward(input_ids, positions, kv_caches, attn_metadata)``
    for a prefill and for a decode step (the fused fast path), ``compute_logits`` on the selected rows.  Same results as
    this package's own load_model on the same checkpoint (whose linears are oracle-checked above)."""
    import types
    from aphrodite_engine_amd.reference_model import MI355XLlamaForCausalLM
    CU.write_checkpoint(str(tmp_path), CFG, "gptq", seed=21)
    hf_dict = L.read_hf_config(str(tmp_path))
    hf = types.SimpleNamespace(**hf_dict)
    hf.to_dict = lambda: dict(hf_dict)
    hf.torch_dtype = torch.float16
    qc = L.resolve_quant_config(str(tmp_path), hf_dict, None, torch.float16)
    with torch.no_grad():
        ref = L.load_model(str(tmp_path), dtype=torch.float16, device=DEV)
        m = MI355XLlamaForCausalLM(config=hf, cache_config=types.SimpleNamespace(cache_dtype="auto"), quant_config=qc)
        m.load_weights(L.iter_safetensors(str(tmp_path)))
        m.to(DEV)
        for _, module in m.named_modules():                   # DefaultModelLoader.load_model's post-load pass
            qm = getattr(module, "quant_method", None)
            if qm is not None:
                qm.process_weights_after_loading(module)
        # prefill of 3 prompts, then one decode step, through both models
        from aphrodite_engine_amd.attention.backend import MI355XAttentionMetadata   # noqa: F401  (the metadata both accept)
        lens = [5, 17, 33]
        bs, block = len(lens), 16
        meta_d, pos_d, nblocks = M.make_decode_metadata(bs, lens, block, DEV)
        ids = torch.randint(0, CFG.vocab_size, (bs, ), device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
        outs = []
        for model in (ref, m):
            caches = M.make_kv_caches(CFG, nblocks, block, torch.float16, "auto", DEV, seed=9)
            hidden = model(ids, pos_d, caches, meta_d) if model is ref else model.forward(ids, pos_d, caches, meta_d, None)
            outs.append(hidden)
        assert m.inner.use_fused_decode and all(l.fused_decode_ok(bs) for l in m.inner.layers)
        assert torch.isfinite(outs[1].float()).all()
        torch.testing.assert_close(outs[1].float(), outs[0].float(), rtol=2e-3, atol=2e-3)
        sm = types.SimpleNamespace(selected_token_indices=torch.tensor([0, 2], device=DEV))
        logits = m.compute_logits(outs[1], sm)
        assert logits.shape == (2, CFG.vocab_size)
        torch.testing.assert_close(logits.float(), ref.compute_logits(outs[0][[0, 2]]).float(), rtol=2e-3, atol=2e-3)
        # a second post-load pass (the adapter's own _finish already ran one) must not repack again
        before = m.inner.layers[0].qkv_proj.qweight.clone()
        m.inner.process_weights_after_loading()
        assert torch.equal(before, m.inner.layers[0].qkv_proj.qweight)
        with pytest.raises(NotImplementedError):
            m.forward(ids, pos_d, caches, meta_d, intermediate_tensors=object())
