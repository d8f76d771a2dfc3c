#!/usr/bin/env python3
"""Generate golden input/output vectors from the REFERENCE's own Python.

Run in the build container only (needs /root/reference; the GPU box does not
have it):   python tests/golden/make_golden.py

The reference package is not importable as a whole here (missing loguru,
msgspec, ...), so this script loads the few pure-Python files it needs *by
path* under stub parent packages, and lifts the torch reference functions out
of the reference's test files with ``ast`` (source text is executed from where
it lies; nothing is copied into this repo).  Outputs: small ``.npz`` fixtures
under tests/golden/, committed, which pin ``oracle/`` (tests/test_oracle_golden.py).

Sources exercised:
  aphrodite/quantization/utils/quant_utils.py   quantize_weights, gptq_pack,
      awq_pack, pack_cols, unpack_cols, sort_weights, permute_rows
  tests/kernels/test_awq_triton.py              awq_dequantize_torch
  tests/kernels/quant_utils.py                  ref_dynamic_per_token_quant,
                                                ref_dynamic_per_tensor_fp8_quant
  tests/kernels/test_cutlass.py                 baseline_scaled_mm
  tests/kernels/test_attention.py               ref_single_query_cached_kv_attention
  tests/kernels/test_cache.py                   (scatter reference, restated
                                                 inline from :176-192)
  aphrodite/modeling/layers/sampler.py          _apply_top_k_top_p, _multinomial
  aphrodite/quantization/compressed_tensors/utils.py   should_ignore_layer, QuantizationArgs
  aphrodite/quantization/compressed_tensors/compressed_tensors.py   _get_scheme_from_parts + predicates
  aphrodite/modeling/layers/linear.py           MergedColumnParallelLinear / QKVParallelLinear /
                                                RowParallelLinear .weight_loader (methods lifted out of
                                                their classes, run with a stand-in self)
  tests/kernels/test_flash_attn.py              ref_paged_attn (prefill over a paged cache, sliding window)
  tests/kernels/test_moe.py                     torch_moe (+ SiluAndMul.forward_native)
  aphrodite/modeling/layers/rotary_embedding.py RotaryEmbedding._compute_cos_sin_cache, Llama3RotaryEmbedding
  aphrodite/quantization/{gptq,awq,fp8}.py, compressed_tensors/schemes/compressed_tensors_w8a8_fp8.py
                                                *.create_weights (parameter shapes, dtypes, loader metadata)
  aphrodite/quantization/kv_cache.py            BaseKVCacheMethod.process_weights_after_loading
  aphrodite/modeling/model_loader/weight_utils.py  kv_cache_scales_loader (+ quantization/schema.py)
"""
import ast
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("APHRODITE_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []
    sys.modules[name] = m
    return m


def _load(name, relpath):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def _lift(relpath, names, extra_globals):
    """exec the named top-level functions/assignments of a reference file."""
    src = open(os.path.join(REF, relpath)).read()
    tree = ast.parse(src)
    ns = dict(extra_globals)
    for node in tree.body:
        tgt = None
        if isinstance(node, ast.FunctionDef) and node.name in names:
            tgt = node
        elif isinstance(node, ast.Assign) and any(
                isinstance(t, ast.Name) and t.id in names for t in node.targets):
            tgt = node
        if tgt is not None:
            seg = ast.get_source_segment(src, tgt).replace("'cuda'", "'cpu'")
            exec(compile(seg, relpath, "exec"), ns)
    return ns


def load_reference():
    class _Logger:
        def __getattr__(self, _):
            return lambda *a, **k: None
    _stub("loguru", logger=_Logger())
    _stub("aphrodite")
    _stub("aphrodite.quantization")
    _stub("aphrodite.quantization.utils")
    _stub("aphrodite.quantization.qqq", MARLIN_QQQ_SUPPORTED_NUM_BITS=[4])
    # The reference's pure-Python ScalarType (aphrodite/_core_ext.py:28-171) is
    # a typing mock (min()/max() raise, uint() mis-sets the exponent); the
    # real one is C++ (kernels/core/scalar_type.hpp:12-260), which is not built
    # here.  Stand in a minimal integer-only type with the C++ semantics:
    # value = stored - bias, min = (signed ? -2^(bits-1) : 0) - bias.
    class ScalarType:
        def __init__(self, size_bits, bias=0, signed=False):
            self.size_bits, self.bias, self.signed = size_bits, bias, signed
        def is_integer(self): return True
        def is_signed(self): return self.signed
        def has_bias(self): return self.bias != 0
        def min(self):
            return (-(1 << (self.size_bits - 1)) if self.signed else 0) - self.bias
        def max(self):
            return ((1 << (self.size_bits - (1 if self.signed else 0))) - 1) - self.bias
        def __eq__(self, o):
            return (self.size_bits, self.bias, self.signed) == (o.size_bits, o.bias, o.signed)
        def __hash__(self):
            return hash((self.size_bits, self.bias, self.signed))
    class scalar_types:
        uint4 = ScalarType(4)
        uint8 = ScalarType(8)
        uint4b8 = ScalarType(4, 8)
        uint8b128 = ScalarType(8, 128)
    st = _stub("aphrodite.scalar_type", ScalarType=ScalarType,
               scalar_types=scalar_types)
    qu = _load("aphrodite.quantization.utils.quant_utils",
               "aphrodite/quantization/utils/quant_utils.py")
    return st, qu


def main():
    st, qu = load_reference()
    from typing import List, Optional, Tuple, Type, Union
    g = dict(torch=torch, Optional=Optional, List=List, Tuple=Tuple,
             Union=Union, Type=Type)

    # ---------------- int4 formats --------------------------------------
    torch.manual_seed(0)
    K, N, G = 256, 64, 128
    w = (torch.randn(K, N) * 0.02).half()
    out = {}
    w_ref, w_q, w_s, _ = qu.quantize_weights(w, st.scalar_types.uint4b8, G)
    out.update(w=w.numpy(), sym_w_ref=w_ref.numpy(), sym_w_q=w_q.numpy(),
               sym_w_s=w_s.numpy())
    out["gptq_packed"] = qu.gptq_pack(w_q, 4, K, N).numpy()
    w_ref2, w_q2, w_s2, w_zp2 = qu.quantize_weights(
        w, st.scalar_types.uint4, G, zero_points=True)
    out.update(zp_w_ref=w_ref2.numpy(), zp_w_q=w_q2.numpy(),
               zp_w_s=w_s2.numpy(), zp_w_zp=w_zp2.numpy())
    out["awq_packed"] = qu.awq_pack(w_q2, 4, K, N).numpy()
    out["awq_zeros_packed"] = qu.awq_pack(w_zp2, 4, K // G, N).numpy()
    out["pack_cols"] = qu.pack_cols(w_q2, 4, K, N).numpy()
    out["unpack_cols"] = qu.unpack_cols(qu.pack_cols(w_q2, 4, K, N), 4, K, N).numpy()
    # act-order helpers
    torch.manual_seed(1)
    w_ref3, w_q3, g_idx3, rand_perm3 = qu.permute_rows(w_q.clone(), w_ref.clone(), G)
    q_sorted, g_sorted, sort_idx = qu.sort_weights(w_q3, g_idx3)
    out.update(act_w_q=w_q3.numpy(), act_g_idx=g_idx3.numpy(),
               act_rand_perm=rand_perm3.numpy(), act_sorted_q=q_sorted.numpy(),
               act_sorted_g=g_sorted.numpy(), act_sort_idx=sort_idx.numpy(),
               act_w_ref=w_ref3.numpy())
    # AWQ dequant torch oracle of the reference test
    ns = _lift("tests/kernels/test_awq_triton.py",
               {"reverse_awq_order", "awq_dequantize_torch"}, g)
    awq_dq = ns["awq_dequantize_torch"](
        torch.from_numpy(out["awq_packed"]), w_s2,
        torch.from_numpy(out["awq_zeros_packed"]), G)
    out["awq_dequant"] = awq_dq.to(torch.float16).numpy()
    np.savez_compressed(os.path.join(OUT, "int4_formats.npz"), **out)

    # ---------------- fp8 quant + scaled mm -------------------------------
    _stub("aphrodite.common")
    _stub("aphrodite.common.utils", is_hip=lambda: False)
    ns = _lift("tests/kernels/quant_utils.py",
               {"ROCM_FP8_MAX", "FP8_DTYPE", "as_float32_tensor",
                "ref_dynamic_per_token_quant",
                "ref_dynamic_per_tensor_fp8_quant"},
               dict(g, is_hip=lambda: False))
    torch.manual_seed(2)
    x = (torch.rand(7, 96) - 0.5) * 40
    x[3] *= 1e-4                      # hits the min-scale floor
    xh = x.half()
    q_tok, s_tok = ns["ref_dynamic_per_token_quant"](xh, torch.float8_e4m3fn)
    ub = torch.tensor([5.0])
    q_ub, s_ub = ns["ref_dynamic_per_token_quant"](xh, torch.float8_e4m3fn, ub)
    q_ten, s_ten = ns["ref_dynamic_per_tensor_fp8_quant"](xh)
    f8 = {}
    f8.update(x=xh.numpy(), q_tok=q_tok.view(torch.uint8).numpy(),
              s_tok=s_tok.numpy(), q_ub=q_ub.view(torch.uint8).numpy(),
              s_ub=s_ub.numpy(), ub=ub.numpy(),
              q_ten=q_ten.view(torch.uint8).numpy(),
              s_ten=np.asarray(s_ten.numpy()).reshape(1))
    ns2 = _lift("tests/kernels/test_cutlass.py", {"baseline_scaled_mm"}, g)
    torch.manual_seed(3)
    a = (torch.randn(5, 64) * 2).to(torch.float8_e4m3fn)
    b = (torch.randn(64, 48) * 2).to(torch.float8_e4m3fn)
    sa = torch.rand(5, 1) + 0.1
    sb = torch.rand(1, 48) + 0.1
    bias = torch.randn(48)
    mm = ns2["baseline_scaled_mm"](a, b, sa, sb, torch.float32, bias)
    f8.update(mm_a=a.view(torch.uint8).numpy(), mm_b=b.view(torch.uint8).numpy(),
              mm_sa=sa.numpy(), mm_sb=sb.numpy(), mm_bias=bias.numpy(),
              mm_out=mm.numpy())
    # codec KATs: every e4m3fn / e5m2 byte through torch's own decode
    allb = torch.arange(256, dtype=torch.uint8)
    f8["e4m3_table"] = allb.view(torch.float8_e4m3fn).float().numpy()
    f8["e5m2_table"] = allb.view(torch.float8_e5m2).float().numpy()
    np.savez_compressed(os.path.join(OUT, "fp8.npz"), **f8)

    # ---------------- paged attention + cache write -----------------------
    ns3 = _lift("tests/kernels/test_attention.py",
                {"ref_masked_attention", "ref_single_query_cached_kv_attention"}, g)
    torch.manual_seed(4)
    S, Hq, Hkv, D, BS, NB = 4, 8, 2, 64, 16, 12
    x_ = 4  # float32 cache -> x = 16/4
    q = torch.randn(S, Hq, D) * 0.5
    kc = (torch.rand(NB, Hkv, D // x_, BS, x_) - 0.5) * 2 * D ** -0.5 * 4
    vc = (torch.rand(NB, Hkv, D, BS) - 0.5) * 2 * D ** -0.5 * 4
    seq_lens = torch.tensor([1, 16, 37, 150], dtype=torch.int32)
    maxb = (150 + BS - 1) // BS
    bt = torch.stack([torch.randperm(NB)[:maxb] for _ in range(S)]).int()
    slopes = torch.randn(Hq)
    att = {}
    for tag, al in (("plain", None), ("alibi", slopes)):
        o = torch.empty(S, Hq, D)
        ns3["ref_single_query_cached_kv_attention"](
            o, q, Hq // Hkv, kc, vc, bt, seq_lens, 0.125, al)
        att["out_" + tag] = o.numpy()
    att.update(q=q.numpy(), kc=kc.numpy(), vc=vc.numpy(), bt=bt.numpy(),
               seq_lens=seq_lens.numpy(), slopes=slopes.numpy())
    # reshape_and_cache reference loop (tests/kernels/test_cache.py:176-192)
    torch.manual_seed(5)
    T = 9
    key = torch.randn(T, Hkv, D)
    val = torch.randn(T, Hkv, D)
    slots = torch.tensor(np.random.RandomState(5).choice(NB * BS, T, replace=False))
    ckc, cvc = kc.clone(), vc.clone()
    rk = key.reshape(T, *ckc[0, :, :, 0, :].shape)
    for i in range(T):
        bi, bo = int(slots[i]) // BS, int(slots[i]) % BS
        ckc[bi, :, :, bo, :] = rk[i]
        cvc[bi, :, :, bo] = val[i]
    att.update(rc_key=key.numpy(), rc_val=val.numpy(), rc_slots=slots.numpy(),
               rc_kc=ckc.numpy(), rc_vc=cvc.numpy())
    np.savez_compressed(os.path.join(OUT, "attention.npz"), **att)

    # ---------------- prefill over a paged cache: tests/kernels/test_flash_attn.py ref_paged_attn (:17-70) -------
    ns_fa = _lift("tests/kernels/test_flash_attn.py", {"ref_paged_attn"}, g)
    torch.manual_seed(17)
    fa_hq, fa_hkv, fa_hd, fa_bs, fa_nb = 4, 2, 32, 16, 12
    q_lens, kv_lens = [5, 1, 17, 9], [13, 40, 17, 9]          # the last two: no cached context (plain causal prefill)
    fq = torch.randn(sum(q_lens), fa_hq, fa_hd)
    fkc = torch.randn(fa_nb, fa_bs, fa_hkv, fa_hd)            # flash layout [blocks, block, heads, dim]
    fvc = torch.randn(fa_nb, fa_bs, fa_hkv, fa_hd)
    fbt = torch.stack([torch.randperm(fa_nb)[:3] for _ in q_lens]).int()
    fa = dict(q=fq.numpy(), kc=fkc.numpy(), vc=fvc.numpy(), bt=fbt.numpy(), q_lens=np.array(q_lens),
              kv_lens=np.array(kv_lens), scale=np.float32(fa_hd ** -0.5))
    for tag, win in (("full", None), ("win8", 8)):
        fa["out_" + tag] = ns_fa["ref_paged_attn"](fq.clone(), fkc, fvc, q_lens, kv_lens, fbt, fa_hd ** -0.5,
                                                   sliding_window=win).numpy()
    np.savez_compressed(os.path.join(OUT, "prefill_paged.npz"), **fa)

    # ---------------- sampler: top-k / top-p masking and the multinomial draw --------------------
    # aphrodite/modeling/layers/sampler.py:865-891 and :1273-1292, lifted as they lie
    ns4 = _lift("aphrodite/modeling/layers/sampler.py", {"_apply_top_k_top_p", "_multinomial", "_apply_min_p"},
                dict(g, SequenceGroupToSample=object))
    torch.manual_seed(11)
    B, V = 7, 4096
    logits = torch.randn(B, V) * 3
    logits[6] = torch.round(logits[6] * 4) / 4                 # a row with ties (kept set may differ in ties)
    top_k = torch.tensor([50, V, 1, 40, V, 7, 25])
    top_p = torch.tensor([0.9, 0.8, 1.0, 0.95, 0.5, 1.0, 0.7])
    masked = ns4["_apply_top_k_top_p"](logits.clone(), top_p, top_k)
    probs = torch.softmax(masked, dim=-1, dtype=torch.float)
    torch.manual_seed(12)
    q = torch.empty_like(probs).exponential_()                 # the draws _multinomial makes with this seed
    torch.manual_seed(12)
    ids = ns4["_multinomial"](probs.clone(), 1).reshape(-1)
    min_p = torch.tensor([0.05, 0.0, 0.2, 0.01, 0.5, 0.1, 0.02])
    masked_minp = ns4["_apply_min_p"](masked.clone(), min_p.clone())
    np.savez_compressed(os.path.join(OUT, "sampler.npz"), logits=logits.numpy(), top_k=top_k.numpy(),
                        top_p=top_p.numpy(), masked=masked.numpy(), probs=probs.numpy(), q=q.numpy(),
                        ids=ids.numpy(), min_p=min_p.numpy(), masked_minp=masked_minp.numpy())

    # ---------------- compressed-tensors: which layers a checkpoint left unquantised ----------------
    # compressed_tensors/utils.py:113-171, 242-260 (should_ignore_layer and helpers)
    import json
    import re
    ns5 = _lift("aphrodite/quantization/compressed_tensors/utils.py",
                {"should_ignore_layer", "check_equal_or_regex_match", "_is_equal_or_regex_match"},
                dict(g, re=re, Iterable=__import__("typing").Iterable,
                     FUSED_LAYER_NAME_MAPPING=qu.FUSED_LAYER_NAME_MAPPING))
    cases = []
    ignores = [["lm_head"], ["re:.*self_attn.*"], ["model.layers.0.mlp.gate_proj", "model.layers.0.mlp.up_proj"],
               ["model.layers.1.self_attn.q_proj"], ["re:model\\.layers\\.[01]\\.mlp\\..*"], []]
    names = ["lm_head", "model.layers.0.self_attn.qkv_proj", "model.layers.0.self_attn.o_proj",
             "model.layers.0.mlp.gate_up_proj", "model.layers.1.mlp.down_proj", "model.layers.1.self_attn.qkv_proj",
             "model.layers.2.mlp.gate_up_proj"]
    for ig in ignores:
        for n in names:
            try:
                res = bool(ns5["should_ignore_layer"](n, ignore=ig))
            except ValueError:
                res = "ValueError"
            cases.append({"layer": n, "ignore": ig, "result": res})
    with open(os.path.join(OUT, "ct_ignore.json"), "w") as f:
        json.dump(cases, f, indent=0)
    # ---------------- tensor-parallel weight loaders: the reference's own methods on small tensors ----
    # modeling/layers/linear.py: MergedColumnParallelLinear.weight_loader (:452-595),
    # QKVParallelLinear.weight_loader (:815-988), RowParallelLinear.weight_loader (:1072-1112), lifted out
    # of their classes and run with a stand-in ``self`` for every (world, rank) below.
    import textwrap
    from torch.nn.parameter import Parameter, UninitializedParameter
    tp = {"rank": 0, "world": 1}
    lg = dict(g, Parameter=Parameter, UninitializedParameter=UninitializedParameter, Dict=__import__("typing").Dict,
              get_tensor_model_parallel_rank=lambda: tp["rank"],
              get_tensor_model_parallel_world_size=lambda: tp["world"], logger=type("L", (), {
                  "warning": staticmethod(lambda *a, **k: None)})())
    lg.update(_lift("aphrodite/modeling/layers/linear.py",
                    {"adjust_marlin_shard", "adjust_bitsandbytes_4bit_shard", "adjust_scalar_to_fused_array"}, lg))

    def lift_method(cls, name):
        src = open(os.path.join(REF, "aphrodite/modeling/layers/linear.py")).read()
        for node in ast.parse(src).body:
            if isinstance(node, ast.ClassDef) and node.name == cls:
                for sub in node.body:
                    if isinstance(sub, ast.FunctionDef) and sub.name == name:
                        seg = textwrap.dedent(ast.get_source_segment(src, sub, padded=True))
                        ns = dict(lg)
                        exec(compile(seg, "linear.py", "exec"), ns)
                        return ns[name]
        raise KeyError((cls, name))

    def holder(cls, **attrs):
        k = type("Ref" + cls, (), {"weight_loader": lift_method(cls, "weight_loader")})
        o = k()
        o.__dict__.update(attrs)
        return o

    HQ, HKV, HD, HID, INTER, GS = 4, 2, 16, 64, 96, 32
    rs = np.random.RandomState(7)
    ld = {}

    def rnd(shape, dtype):
        if dtype == torch.int32:
            return torch.from_numpy(rs.randint(-2 ** 31, 2 ** 31 - 1, size=shape, dtype=np.int64).astype(np.int32))
        return torch.from_numpy(rs.standard_normal(shape).astype(np.float32)).to(dtype)

    def mk(shape, dtype, **attrs):
        prm = Parameter(torch.zeros(shape, dtype=dtype), requires_grad=False)
        for k_, v_ in attrs.items():
            setattr(prm, k_, v_)
        return prm

    def param_table(k_loc, n_loc, nshards, row):
        """name -> (shape on this rank, dtype, loader attributes, checkpoint shape given the full (K, N))"""
        kin = 0 if row else None     # group metadata follows K only when K is cut
        return {
            "gptq.qweight": ((k_loc // 8, n_loc), torch.int32, dict(input_dim=0, output_dim=1, packed_dim=0, pack_factor=8)),
            "gptq.qzeros": ((k_loc // GS, n_loc // 8), torch.int32,
                            dict(output_dim=1, packed_dim=1, pack_factor=8, **({"input_dim": 0} if row else {}))),
            "gptq.scales": ((k_loc // GS, n_loc), torch.float16, dict(output_dim=1, **({"input_dim": 0} if row else {}))),
            "gptq.g_idx": ((k_loc, ), torch.int32, dict(input_dim=0)),
            "awq.qweight": ((k_loc, n_loc // 8), torch.int32, dict(input_dim=0, output_dim=1, packed_dim=1, pack_factor=8)),
            "fp8.weight": ((n_loc, k_loc), torch.float16, dict(input_dim=1, output_dim=0)),
            "fp8.weight_scale": ((nshards, ), torch.float32, dict(needs_scalar_to_array=True)),
            "ct.weight_scale": ((n_loc, 1), torch.float32, dict(output_dim=0)),
        }

    def full_shape(name, k, n):
        return {"gptq.qweight": (k // 8, n), "gptq.qzeros": (k // GS, n // 8), "gptq.scales": (k // GS, n),
                "gptq.g_idx": (k, ), "awq.qweight": (k, n // 8), "fp8.weight": (n, k), "fp8.weight_scale": (),
                "ct.weight_scale": (n, 1)}[name]

    for world, rank in ((1, 0), (2, 1), (4, 3)):
        tp["rank"], tp["world"] = rank, world
        nh = HQ // world
        nkv, rep = (1, world // HKV) if world >= HKV else (HKV // world, 1)
        qkv = holder("QKVParallelLinear", quant_config=object(), num_heads=nh, num_kv_heads=nkv, head_size=HD,
                     num_kv_head_replicas=rep, total_num_heads=HQ, total_num_kv_heads=HKV,
                     output_sizes=[nh * HD * world, nkv * HD * world, nkv * HD * world])
        merged = holder("MergedColumnParallelLinear", quant_config=object(), output_sizes=[INTER, INTER])
        row = holder("RowParallelLinear", quant_config=object(), tp_rank=rank, tp_size=world, input_size=INTER)
        layers = {
            "qkv": (qkv, HID, (nh + 2 * nkv) * HD, [("q", HQ * HD), ("k", HKV * HD), ("v", HKV * HD)], False),
            "merged": (merged, HID, 2 * INTER // world, [(0, INTER), (1, INTER)], False),
            "row": (row, INTER // world, HID, [(None, HID)], True),
        }
        for lname, (obj, k_loc, n_loc, shards, is_row) in layers.items():
            k_full = INTER if is_row else HID
            for pname, (shape, dtype, attrs) in param_table(k_loc, n_loc, len(shards), is_row).items():
                if is_row and pname == "ct.weight_scale":
                    continue
                prm = mk(shape, dtype, **attrs)
                for sid, n_full in shards:
                    key_in = f"in.{lname}.{pname}.{sid}"
                    if key_in not in ld:
                        ld[key_in] = rnd(full_shape(pname, k_full, n_full), dtype).numpy()
                    t_in = torch.from_numpy(np.asarray(ld[key_in]))
                    if is_row:
                        obj.weight_loader(prm, t_in)
                    else:
                        obj.weight_loader(prm, t_in, sid)
                ld[f"out.{lname}.{pname}.w{world}r{rank}"] = prm.data.numpy().copy()
                # the same tensors stored fused on disk (Phi-3 style), through loaded_shard_id = None
                if not is_row and pname in ("fp8.weight", "gptq.qweight", "gptq.scales", "ct.weight_scale"):
                    odim = attrs["output_dim"]
                    fused = torch.cat([torch.from_numpy(np.asarray(ld[f"in.{lname}.{pname}.{sid}"]))
                                       for sid, _ in shards], dim=odim)
                    prm2 = mk(shape, dtype, **attrs)
                    obj.weight_loader(prm2, fused, None)
                    assert torch.equal(prm2.data, prm.data)
    # more head layouts for the qkv loader alone: MHA (8/8), MQA (4/1: the KV head replicated on every rank),
    # GQA with fewer KV heads than ranks (8/2 at world 4 and 8)
    for hq, hkv, worlds in ((8, 8, ((2, 0), (4, 2))), (4, 1, ((2, 1), (4, 3))), (8, 2, ((4, 1), (8, 5)))):
        for world, rank in worlds:
            tp["rank"], tp["world"] = rank, world
            nh = hq // world
            nkv, rep = (1, world // hkv) if world >= hkv else (hkv // world, 1)
            qkv = holder("QKVParallelLinear", quant_config=object(), num_heads=nh, num_kv_heads=nkv, head_size=HD,
                         num_kv_head_replicas=rep, total_num_heads=hq, total_num_kv_heads=hkv,
                         output_sizes=[nh * HD * world, nkv * HD * world, nkv * HD * world])
            n_loc = (nh + 2 * nkv) * HD
            tag = f"h{hq}kv{hkv}"
            for pname in ("gptq.qweight", "gptq.qzeros", "gptq.scales", "fp8.weight", "fp8.weight_scale"):
                shape, dtype, attrs = param_table(HID, n_loc, 3, False)[pname]
                prm = mk(shape, dtype, **attrs)
                for sid, n_full in (("q", hq * HD), ("k", hkv * HD), ("v", hkv * HD)):
                    key_in = f"in.qkv_{tag}.{pname}.{sid}"
                    if key_in not in ld:
                        ld[key_in] = rnd(full_shape(pname, HID, n_full), dtype).numpy()
                    qkv.weight_loader(prm, torch.from_numpy(np.asarray(ld[key_in])), sid)
                ld[f"out.qkv_{tag}.{pname}.w{world}r{rank}"] = prm.data.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "loader_shards.npz"), **ld)

    # ---------------- mixture of experts: the reference's torch_moe (tests/kernels/test_moe.py:15-29) ----
    def lift_class_method(relpath, cls, name, glb):
        src = open(os.path.join(REF, relpath)).read()
        for node in ast.parse(src).body:
            if isinstance(node, ast.ClassDef) and node.name == cls:
                for sub in node.body:
                    if isinstance(sub, ast.FunctionDef) and sub.name == name:
                        ns_ = dict(glb)
                        exec(compile(textwrap.dedent(ast.get_source_segment(src, sub, padded=True)), relpath, "exec"),
                             ns_)
                        return ns_[name]
        raise KeyError((cls, name))
    silu_native = lift_class_method("aphrodite/modeling/layers/activation.py", "SiluAndMul", "forward_native",
                                    dict(g, F=torch.nn.functional))
    SiluAndMul = type("SiluAndMul", (), {"__call__": lambda self, x: silu_native(self, x)})
    ns6 = _lift("tests/kernels/test_moe.py", {"torch_moe"}, dict(g, SiluAndMul=SiluAndMul))
    torch.manual_seed(21)
    Bm, Dm, Nm, Em, Tk = 9, 64, 32, 8, 2
    a_m = torch.randn(Bm, Dm) / 4
    w1_m = torch.randn(Em, 2 * Nm, Dm) / 8         # [E, 2N, K] as fused_moe holds them (gate | up rows)
    w2_m = torch.randn(Em, Dm, Nm) / 8
    score = torch.randn(Bm, Em)
    moe_out = ns6["torch_moe"](a_m, w1_m, w2_m, score, Tk)
    np.savez_compressed(os.path.join(OUT, "moe.npz"), a=a_m.numpy(), w1=w1_m.numpy(), w2=w2_m.numpy(),
                        score=score.numpy(), out=moe_out.numpy(), topk=np.int64(Tk))
    # FusedMoE.make_expert_params_mapping (fused_moe/layer.py:455-471): the rows a model's load_weights walks
    mapping = lift_class_method("aphrodite/modeling/layers/fused_moe/layer.py", "FusedMoE", "make_expert_params_mapping", g)
    with open(os.path.join(OUT, "moe_mapping.json"), "w") as f:
        json.dump([list(r) for r in mapping(None, "w1", "w2", "w3", 3)] +
                  [list(r) for r in mapping(None, "gate_proj", "down_proj", "up_proj", 2)], f)

    # ---------------- compressed-tensors: which scheme a config group selects ----------------------------
    # compressed_tensors.py:133-253 (_is_* predicates + _get_scheme_from_parts) on the reference's own pydantic
    # QuantizationArgs; the scheme classes are recorded, not constructed.
    _stub("aphrodite.quantization.compressed_tensors")
    ctu = _load("aphrodite.quantization.compressed_tensors.utils", "aphrodite/quantization/compressed_tensors/utils.py")

    def recorder(name):
        def make(**kw):
            return {"scheme": name, **{k_: (v_.value if hasattr(v_, "value") else v_) for k_, v_ in kw.items()}}
        make.get_min_capability = staticmethod(lambda: 89)
        return make
    ct_glb = dict(g, BaseModel=object, QuantizationStrategy=ctu.QuantizationStrategy, QuantizationType=ctu.QuantizationType,
                  CompressionFormat=ctu.CompressionFormat,
                  is_activation_quantization_format=ctu.is_activation_quantization_format,
                  W4A16SPARSE24_SUPPORTED_BITS=[4], WNA16_SUPPORTED_BITS=[4, 8],
                  CompressedTensorsW4A16Sparse24=recorder("W4A16Sparse24"), CompressedTensorsWNA16=recorder("WNA16"),
                  CompressedTensorsW8A8Fp8=recorder("W8A8Fp8"), CompressedTensorsW8A16Fp8=recorder("W8A16Fp8"),
                  CompressedTensorsW8A8Int8=recorder("W8A8Int8"))
    csrc = "aphrodite/quantization/compressed_tensors/compressed_tensors.py"
    meths = {n: lift_class_method(csrc, "CompressedTensorsConfig", n, ct_glb) for n in
             ("_is_static_tensor_w8a8", "_is_dynamic_token_w8a8", "_is_fp8_w8a8", "_is_fp8_w8a16",
              "_is_wNa16_group_channel", "_get_scheme_from_parts")}
    f8w = {"num_bits": 8, "type": "float", "symmetric": True, "dynamic": False}
    i4w = {"num_bits": 4, "type": "int", "symmetric": True, "dynamic": False}
    groups = [
        ("float-quantized", {**f8w, "strategy": "channel"}, {"num_bits": 8, "type": "float", "strategy": "token", "dynamic": True}),
        ("float-quantized", {**f8w, "strategy": "tensor"}, {"num_bits": 8, "type": "float", "strategy": "tensor", "dynamic": False}),
        ("float-quantized", {**f8w, "strategy": "tensor"}, {"num_bits": 8, "type": "float", "strategy": "token", "dynamic": True}),
        ("float-quantized", {**f8w, "strategy": "channel"}, None),
        ("naive-quantized", {**f8w, "strategy": "tensor"}, None),
        ("pack-quantized", {**i4w, "strategy": "group", "group_size": 128}, None),
        ("pack-quantized", {**i4w, "strategy": "channel"}, None),
        ("pack-quantized", {**i4w, "strategy": "group", "group_size": 128, "actorder": "group"}, None),
        ("pack-quantized", {**i4w, "strategy": "group", "group_size": 128, "actorder": "weight"}, None),
        ("pack-quantized", {**i4w, "num_bits": 8, "strategy": "group", "group_size": 128}, None),
        ("marlin-24", {**i4w, "strategy": "group", "group_size": 128}, None),
        ("int-quantized", {"num_bits": 8, "type": "int", "symmetric": True, "strategy": "channel", "dynamic": False},
         {"num_bits": 8, "type": "int", "symmetric": True, "strategy": "token", "dynamic": True}),
        ("int-quantized", {"num_bits": 8, "type": "int", "symmetric": True, "strategy": "tensor", "dynamic": False},
         {"num_bits": 8, "type": "int", "symmetric": True, "strategy": "tensor", "dynamic": False}),
        ("pack-quantized", {**i4w, "symmetric": False, "strategy": "group", "group_size": 128}, None),
        ("float-quantized", {**f8w, "strategy": "group", "group_size": 128}, {"num_bits": 8, "type": "float", "strategy": "token", "dynamic": True}),
    ]
    scheme_cases = []
    for fmt_, w_, a_ in groups:
        cfg_ = types.SimpleNamespace(quant_format=fmt_, _check_scheme_supported=lambda *x, **k: True)
        for n_, f_ in meths.items():
            setattr(cfg_, n_, types.MethodType(f_, cfg_))
        wq = ctu.QuantizationArgs.parse_obj(w_)
        aq = ctu.QuantizationArgs.parse_obj(a_) if a_ is not None else None
        try:
            res = cfg_._get_scheme_from_parts(wq, aq)
        except NotImplementedError:
            res = "NotImplementedError"
        scheme_cases.append({"format": fmt_, "weights": w_, "input_activations": a_, "result": res})
    with open(os.path.join(OUT, "ct_schemes.json"), "w") as f:
        json.dump(scheme_cases, f, indent=0, default=str)

    # ---------------- parameter contracts: what each quant method registers on a layer ---------------------
    # create_weights of GPTQLinearMethod (gptq.py:102-213), AWQLinearMethod (awq.py:81-139), Fp8LinearMethod
    # (fp8.py:128-180) and the CompressedTensorsW8A8Fp8 scheme (:62-107), run with recording parameter classes.
    from fractions import Fraction

    def rec_param(kind):
        def make(data, weight_loader=None, **kw):
            data._rec = {"kind": kind, **{k_: (int(v_) if isinstance(v_, Fraction) else v_) for k_, v_ in kw.items()}}
            return data
        return make
    pkinds = ("PackedAphroditeParameter", "RowAphroditeParameter", "ChannelQuantScaleParameter",
              "GroupQuantScaleParameter", "PackedColumnParameter", "ModelWeightParameter", "PerTensorScaleParameter",
              "BaseAphroditeParameter")
    pg = dict(g, Callable=__import__("typing").Callable, **{k_: rec_param(k_) for k_ in pkinds})

    class FakeLayer:
        def __init__(self):
            self.params = {}
        def register_parameter(self, name, prm):
            self.params[name] = None if prm is None else {"shape": list(prm.shape), "dtype": str(prm.dtype),
                                                          **prm._rec}

    def lift_class(relpath, cls, glb):
        src = open(os.path.join(REF, relpath)).read()
        for node in ast.parse(src).body:
            if isinstance(node, ast.ClassDef) and node.name == cls:
                ns_ = dict(glb)
                exec(compile(ast.get_source_segment(src, node), relpath, "exec"), ns_)
                return ns_[cls]
        raise KeyError(cls)
    import enum
    ExllamaState = lift_class("aphrodite/quantization/gptq.py", "ExllamaState", dict(g, Enum=enum.Enum, enum=enum))
    gptq_cw = lift_class_method("aphrodite/quantization/gptq.py", "GPTQLinearMethod", "create_weights",
                                dict(pg, ExllamaState=ExllamaState))
    awq_cw = lift_class_method("aphrodite/quantization/awq.py", "AWQLinearMethod", "create_weights", pg)
    fp8_cw = lift_class_method("aphrodite/quantization/fp8.py", "Fp8LinearMethod", "create_weights", pg)
    ctf_cw = lift_class_method("aphrodite/quantization/compressed_tensors/schemes/compressed_tensors_w8a8_fp8.py",
                               "CompressedTensorsW8A8Fp8", "create_weights",
                               dict(pg, QuantizationStrategy=ctu.QuantizationStrategy))
    # (in_per_partition, out_partition_sizes, in_full, out_full)
    geoms = {"column_tp1": (256, [256, 64, 64], 256, 384), "column_tp2": (256, [128, 32, 32], 256, 384),
             "row_tp1": (512, [256], 512, 256), "row_tp2": (256, [256], 512, 256)}
    contracts = []
    for gname, (kin, outs, kfull, nfull) in geoms.items():
        for gs_, desc in ((128, False), (128, True), (-1, False)):
            qc_ = types.SimpleNamespace(group_size=gs_, desc_act=desc, pack_factor=Fraction(32, 4), weight_bits=4)
            lay = FakeLayer()
            gptq_cw(types.SimpleNamespace(quant_config=qc_), lay, kin, outs, kfull, nfull, torch.float16)
            contracts.append({"method": "gptq", "geom": gname, "args": [kin, outs, kfull, nfull],
                              "config": {"group_size": gs_, "desc_act": desc}, "params": lay.params,
                              "exllama_state": lay.exllama_state.name})
        qc_ = types.SimpleNamespace(group_size=128, pack_factor=8, weight_bits=4)
        lay = FakeLayer()
        awq_cw(types.SimpleNamespace(quant_config=qc_), lay, kin, outs, kfull, nfull, torch.float16)
        contracts.append({"method": "awq", "geom": gname, "args": [kin, outs, kfull, nfull],
                          "config": {"group_size": 128}, "params": lay.params})
        for ser, scheme in ((True, "dynamic"), (True, "static"), (False, "dynamic")):
            qc_ = types.SimpleNamespace(is_checkpoint_fp8_serialized=ser, activation_scheme=scheme)
            lay = FakeLayer()
            fp8_cw(types.SimpleNamespace(quant_config=qc_), lay, kin, outs, kfull, nfull, torch.bfloat16)
            contracts.append({"method": "fp8", "geom": gname, "args": [kin, outs, kfull, nfull],
                              "config": {"serialized": ser, "activation_scheme": scheme}, "params": lay.params})
        for strat in ("channel", "tensor"):
            for static in (False, True):
                lay = FakeLayer()
                ctf_cw(types.SimpleNamespace(strategy=ctu.QuantizationStrategy(strat), is_static_input_scheme=static),
                       lay, output_partition_sizes=outs, input_size_per_partition=kin, params_dtype=torch.bfloat16,
                       weight_loader=None)
                contracts.append({"method": "ct-w8a8-fp8", "geom": gname, "args": [kin, outs, kfull, nfull],
                                  "config": {"strategy": strat, "static": static}, "params": lay.params})
    with open(os.path.join(OUT, "param_contracts.json"), "w") as f:
        json.dump(contracts, f, indent=0)

    # ---------------- FP8: fused shards with their own scales -> one scale (w8a8_utils.py:23-28, 54-80) --------
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from oracle import fp8 as of8_

    class _Ops:      # the one custom op the function calls, served by the restatement the fp8 goldens above pin
        @staticmethod
        def scaled_fp8_quant(x, scale):
            bits = of8_.static_scaled_fp8_quant(x.float().numpy(), float(scale))
            return torch.from_numpy(bits).view(torch.float8_e4m3fn), scale
    ns8 = _lift("aphrodite/quantization/utils/w8a8_utils.py", {"per_tensor_dequantize", "requantize_with_max_scale"},
                dict(g, ops=_Ops))
    torch.manual_seed(33)
    widths = [48, 16, 16]
    wq_ = (torch.randn(sum(widths), 64) * 60).clamp(-448, 448).to(torch.float8_e4m3fn)
    ws_ = torch.tensor([0.011, 0.0042, 0.0087])
    mx_, wq_out = ns8["requantize_with_max_scale"](wq_.clone(), ws_.clone(), widths)
    fused_ws = torch.tensor([0.011, torch.finfo(torch.float32).min, torch.finfo(torch.float32).min])   # fused on disk
    mx2_, wq_out2 = ns8["requantize_with_max_scale"](wq_.clone(), fused_ws, widths)
    np.savez_compressed(os.path.join(OUT, "fp8_requant.npz"), w=wq_.view(torch.uint8).numpy(), scales=ws_.numpy(),
                        widths=np.array(widths), out=wq_out.view(torch.uint8).numpy(), max_scale=mx_.numpy(),
                        fused_scales=fused_ws.numpy(), out_fused=wq_out2.view(torch.uint8).numpy(),
                        max_scale_fused=mx2_.numpy())

    # ---------------- rotary tables, plain and Llama-3.1 scaled (rotary_embedding.py:101-120, 680-723) -----
    import math
    rsrc = "aphrodite/modeling/layers/rotary_embedding.py"
    base_inv = lift_class_method(rsrc, "RotaryEmbedding", "_compute_inv_freq", dict(g, math=math))
    base_cache = lift_class_method(rsrc, "RotaryEmbedding", "_compute_cos_sin_cache", dict(g, math=math))
    # the subclass method calls super()._compute_inv_freq(base): bind that one call to the lifted base method
    src_r = open(os.path.join(REF, rsrc)).read()
    l3 = None
    for node in ast.parse(src_r).body:
        if isinstance(node, ast.ClassDef) and node.name == "Llama3RotaryEmbedding":
            for sub in node.body:
                if isinstance(sub, ast.FunctionDef) and sub.name == "_compute_inv_freq":
                    seg = textwrap.dedent(ast.get_source_segment(src_r, sub, padded=True))
                    seg = seg.replace("super()._compute_inv_freq(base)", "_base_inv(self, base)")
                    ns_r = dict(g, math=math, _base_inv=base_inv)
                    exec(compile(seg, rsrc, "exec"), ns_r)
                    l3 = ns_r["_compute_inv_freq"]
    plain = types.SimpleNamespace(rotary_dim=128, base=500000.0, max_position_embeddings=640)
    plain._compute_inv_freq = lambda b: base_inv(plain, b)
    scaled = types.SimpleNamespace(rotary_dim=128, base=500000.0, max_position_embeddings=640, scaling_factor=8.0,
                                   low_freq_factor=1.0, high_freq_factor=4.0, orig_max_position=8192)
    scaled._compute_inv_freq = lambda b: l3(scaled, b)
    np.savez_compressed(os.path.join(OUT, "rope.npz"), plain=base_cache(plain).numpy(),
                        llama3=base_cache(scaled).numpy())

    # ---------------- FP8 KV-cache scales: checkpoint rule and the quantization_param_path json ---------
    kv_rule = lift_class_method("aphrodite/quantization/kv_cache.py", "BaseKVCacheMethod",
                                "process_weights_after_loading", dict(g, print_warning_once=lambda *a, **k: None))
    kv_cases = []
    for kvd in ("auto", "fp8", "fp8_e5m2"):
        for ks, vs in ((-1.0, -1.0), (0.02, 0.03), (0.05, -1.0), (1.0, 1.0)):
            layer = types.SimpleNamespace(kv_cache_dtype=kvd, k_scale=torch.tensor(ks), v_scale=torch.tensor(vs))
            kv_rule(None, layer)
            kv_cases.append({"kv_cache_dtype": kvd, "k_scale": ks, "v_scale": vs,
                             "out": [getattr(layer, "_k_scale", None), getattr(layer, "_v_scale", None)]})
    _stub("aphrodite.quantization.schema")
    schema = _load("aphrodite.quantization.schema", "aphrodite/quantization/schema.py")
    ns7 = _lift("aphrodite/modeling/model_loader/weight_utils.py", {"kv_cache_scales_loader"},
                dict(g, json=json, Iterable=__import__("typing").Iterable, QuantParamSchema=schema.QuantParamSchema,
                     logger=type("L", (), {"error": staticmethod(lambda *a, **k: None),
                                           "warning": staticmethod(lambda *a, **k: None)})()))
    import tempfile
    doc = {"model_type": "llama", "kv_cache": {"dtype": "float8_e4m3fn", "scaling_factor": {
        "0": {"0": 0.1, "1": 0.2, "2": 0.3}, "1": {"0": 0.4, "1": 0.5, "2": 0.6}}}}
    json_cases = []
    variants = [("ok", doc, dict(tp_rank=1, tp_size=2, layers=3, model_type="llama")),
                ("wrong_tp", doc, dict(tp_rank=0, tp_size=1, layers=3, model_type="llama")),
                ("wrong_layers", doc, dict(tp_rank=0, tp_size=2, layers=4, model_type="llama")),
                ("wrong_model", doc, dict(tp_rank=0, tp_size=2, layers=3, model_type="mixtral")),
                ("wrong_dtype", {**doc, "kv_cache": {**doc["kv_cache"], "dtype": "float8_e5m2"}},
                 dict(tp_rank=0, tp_size=2, layers=3, model_type="llama"))]
    for tag, d_, kw in variants:
        with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as tf:
            json.dump(d_, tf)
        res = list(ns7["kv_cache_scales_loader"](tf.name, kw["tp_rank"], kw["tp_size"], kw["layers"], kw["model_type"]))
        os.unlink(tf.name)
        json_cases.append({"tag": tag, "doc": d_, "args": kw, "result": [[int(a_), float(b_)] for a_, b_ in res]})
    with open(os.path.join(OUT, "kv_scales.json"), "w") as f:
        json.dump({"rule": kv_cases, "param_path": json_cases}, f, indent=0)
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
