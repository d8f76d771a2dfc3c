"""Mixture-of-experts layer over the MI355X W4A16 kernels -- host mirror of
``aphrodite/modeling/layers/fused_moe/fused_moe.py`` (``fused_topk`` :369-402, ``moe_align_block_size``
:174-228, ``fused_marlin_moe`` :438-542) and of the dense-loop ``MixtralMoE`` the reference actually runs
for GPTQ Mixtral (``modeling/models/mixtral_quant.py:91-156``).

Data flow of ``fused_wna16_moe`` (4 launches after routing, no per-expert loop, every active expert's
weights read once):

    moe_align_block_size(block 16)      sort the (token, k) slots by expert, pad to 16-row blocks
    moe_gather_pack                     packed activations in sorted order
    wna16_gemm_grouped(w13, silu_pack)  [gate_j, up_j interleaved] -> SiluAndMul in the epilogue
    wna16_gemm_grouped(w2, slabs)       fp32 split-K slabs
    moe_combine                         routed weight, sum over top-k
"""
import abc
from typing import Callable, List, Optional, Tuple

import torch
from torch import nn

from .switches import switch
from . import _custom_ops as ops
from .quantization.base_config import QuantizeMethodBase, _param

MOE_BLOCK_M = 16   # one MFMA m-tile = one block of moe_align_block_size


def fused_topk(hidden_states: torch.Tensor, gating_output: torch.Tensor, topk: int,
               renormalize: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    """fused_moe.py:369-402."""
    assert hidden_states.shape[0] == gating_output.shape[0], "Number of tokens mismatch"
    m = hidden_states.shape[0]
    topk_weights = torch.empty(m, topk, dtype=torch.float32, device=hidden_states.device)
    topk_ids = torch.empty(m, topk, dtype=torch.int32, device=hidden_states.device)
    token_expert_indicies = torch.empty(m, topk, dtype=torch.int32, device=hidden_states.device)
    ops.topk_softmax(topk_weights, topk_ids, token_expert_indicies, gating_output.float().contiguous())
    if renormalize:
        topk_weights = topk_weights / topk_weights.sum(dim=-1, keepdim=True)
    return topk_weights, topk_ids


def moe_align_block_size(topk_ids: torch.Tensor, block_size: int, num_experts: int,
                         want_inverse: bool = False):
    """fused_moe.py:174-228 (same outputs; the optional inverse map is ours)."""
    max_num_tokens_padded = topk_ids.numel() + num_experts * (block_size - 1)
    sorted_ids = torch.empty((max_num_tokens_padded, ), dtype=torch.int32, device=topk_ids.device)
    max_num_m_blocks = (max_num_tokens_padded + block_size - 1) // block_size
    expert_ids = torch.empty((max_num_m_blocks, ), dtype=torch.int32, device=topk_ids.device)
    num_tokens_post_pad = torch.empty((1, ), dtype=torch.int32, device=topk_ids.device)
    inv = torch.empty(topk_ids.numel(), dtype=torch.int32, device=topk_ids.device) if want_inverse else None
    ops.moe_align_block_size(topk_ids, num_experts, block_size, sorted_ids, expert_ids, num_tokens_post_pad, inv)
    if want_inverse:
        return sorted_ids, expert_ids, num_tokens_post_pad, inv
    return sorted_ids, expert_ids, num_tokens_post_pad


class DeferredCombine:
    """The second expert GEMM's fp32 slabs + what moe_combine needs, handed to the NEXT norm launch instead of a combine
    launch of its own (ops.fused_add_rms_norm_pack_combine folds it into its input stage: same bits)."""

    def __init__(self, slabs: torch.Tensor, inv: torch.Tensor, topk_weights: torch.Tensor, dtype: torch.dtype):
        self.slabs, self.inv, self.topk_weights, self.dtype = slabs, inv, topk_weights, dtype

    def combine(self) -> torch.Tensor:
        return ops.moe_combine(self.slabs, self.inv, self.topk_weights, self.dtype)


def route_and_align(hidden_states: torch.Tensor, gating_output: torch.Tensor, topk: int, renormalize: bool,
                    num_experts: int, want_inverse: bool = False, custom_routing_function: Optional[Callable] = None):
    """fused_topk + moe_align_block_size(MOE_BLOCK_M): ONE launch for decode-sized batches (ops.moe_route_align), the
    separate ops above otherwise -- same results either way.  Returns (topk_weights, topk_ids, sorted_ids, expert_ids,
    num_tokens_post_pad, inv or None).  ``custom_routing_function`` (FusedMoE.select_experts, layer.py:400-430): the
    model's own routing, called as the reference calls it, then aligned."""
    import os
    t_ = hidden_states.shape[0]
    if custom_routing_function is not None:
        topk_weights, topk_ids = custom_routing_function(hidden_states=hidden_states, gating_output=gating_output,
                                                         topk=topk, renormalize=renormalize)
        topk_weights, topk_ids = topk_weights.float().contiguous(), topk_ids.to(torch.int32).contiguous()
        out = moe_align_block_size(topk_ids, MOE_BLOCK_M, num_experts, want_inverse=want_inverse)
        return (topk_weights, topk_ids) + tuple(out) + (() if want_inverse else (None, ))
    # the one-workgroup launcher's own limits (csrc/moe.hip aphro_moe_route_align): <= 256 experts, topk <= experts, and
    # (slots + 66 E + 1) int32 of LDS <= 64 KB -- e.g. E = 128 with top-8 stops at 991 tokens, E >= 249 is never served.
    # Outside them the separate ops below handle the call (the launcher's check used to surface as a RuntimeError)
    if (0 < t_ * topk <= ops.MOE_ROUTE_ALIGN_MAX_SLOTS and topk <= 8 and topk <= num_experts <= 256
            and (t_ * topk + 66 * num_experts + 1) * 4 <= 64 * 1024
            and gating_output.stride(1) == 1
            and gating_output.dtype in (torch.float16, torch.bfloat16, torch.float32)
            and not switch("APHRO_MOE_NO_ROUTE_ALIGN")):
        return ops.moe_route_align(gating_output, topk, renormalize, num_experts, MOE_BLOCK_M, want_inverse)
    topk_weights, topk_ids = fused_topk(hidden_states, gating_output, topk, renormalize)
    out = moe_align_block_size(topk_ids, MOE_BLOCK_M, num_experts, want_inverse=want_inverse)
    return (topk_weights, topk_ids) + tuple(out) + (() if want_inverse else (None, ))


class Wna16Experts:
    """Stacked int4 expert weights in the layouts the grouped kernel consumes.

    w13: merged [w1 | w3] (gate | up) per expert, K-packed exllama order [E, K/8, 2I] with the gate/up
    columns INTERLEAVED (``ops.interleave_gate_up``) ; w2: [E, I/8, H].  Built from per-expert GPTQ
    tensor sets (qweight [K/8, N] plain GPTQ order, qzeros [G, N/8] stored zero-1, scales [G, N])."""

    def __init__(self, w13_sets, w2_sets, zero_offset: int = 1, layout: str = "gptq"):
        """layout "gptq": tensors as above; "awq": qweight [K, N/8] / qzeros [G, N/8] in the AWQ
        nibble order (zeros stored as-is: zero_offset 0)."""
        empty = torch.empty(0, dtype=torch.int32, device=w13_sets[0][0].device)

        def prep(qw, qz, sc, interleave):
            if layout == "awq":
                k, n = qw.shape[0], qw.shape[1] * 8
                qw = ops.awq_marlin_repack(qw.contiguous(), k, n, 4)
                qz = ops.awq_repack_zeros(qz.contiguous(), n)
            else:
                k8, n = qw.shape
                qw = ops.gptq_marlin_repack(qw.contiguous(), empty, k8 * 8, n, 4)
            if interleave:
                qw, qz, sc = ops.interleave_gate_up(qw, qz, sc)
            return qw, qz.contiguous(), sc.contiguous()

        a = [prep(*s_, True) for s_ in w13_sets]
        b = [prep(*s_, False) for s_ in w2_sets]
        self.w13 = tuple(torch.stack([x[i] for x in a]).contiguous() for i in range(3))
        self.w2 = tuple(torch.stack([x[i] for x in b]).contiguous() for i in range(3))
        self.zero_offset = zero_offset
        self.num_experts = len(w13_sets)
        self.hidden = self.w13[0].shape[1] * 8
        self.inter = self.w2[0].shape[1] * 8


def fused_wna16_moe(hidden_states: torch.Tensor, experts: Wna16Experts, gating_output: torch.Tensor,
                    topk: int, renormalize: bool = True,
                    topk_weights: Optional[torch.Tensor] = None,
                    topk_ids: Optional[torch.Tensor] = None, aligned=None, defer_combine: bool = False):
    """The fused_marlin_moe role (fused_moe.py:438-542) for GPTQ/AWQ int4 experts.  aligned: (sorted_ids, expert_ids,
    post_pad, inv) when the caller has routed and aligned already (route_and_align)."""
    assert hidden_states.shape[1] == experts.hidden, "Hidden size mismatch"
    assert gating_output.shape[1] == experts.num_experts, "Number of experts mismatch"
    m = hidden_states.shape[0]
    e = experts.num_experts
    packed = None
    if aligned is not None:
        sorted_ids, expert_ids, post_pad, inv = aligned
    elif topk_ids is None:
        if (gating_output.dtype == hidden_states.dtype and gating_output.dtype in (torch.float16, torch.bfloat16)
                and gating_output.stride(1) == 1 and hidden_states.stride(1) == 1
                and ops.moe_route_gather_supported(m, e, topk, MOE_BLOCK_M, experts.hidden)
                and not switch("APHRO_MOE_NO_ROUTE_ALIGN")):
            # decode-sized batch, <= 16 experts: routing, alignment and the packed gather in ONE launch (round 6)
            topk_weights, topk_ids, sorted_ids, expert_ids, post_pad, inv, packed, m_pad = ops.moe_route_gather(
                hidden_states, gating_output, topk, renormalize, e, MOE_BLOCK_M)
        else:
            topk_weights, topk_ids, sorted_ids, expert_ids, post_pad, inv = route_and_align(
                hidden_states, gating_output, topk, renormalize, e, want_inverse=True)
    else:
        sorted_ids, expert_ids, post_pad, inv = moe_align_block_size(topk_ids, MOE_BLOCK_M, e, want_inverse=True)
    if packed is None:
        m_pad = (sorted_ids.numel() + MOE_BLOCK_M - 1) // MOE_BLOCK_M * MOE_BLOCK_M
        packed = ops.moe_gather_pack(hidden_states, sorted_ids, post_pad, m_pad, topk)
    qw, qz, sc = experts.w13
    n13 = qw.shape[2]
    if ops.wna16_grouped_ksplit(m_pad, n13, experts.hidden, sc.shape[1]) == 1 and n13 % 256 == 0:
        act = ops.wna16_gemm_grouped(packed, m_pad, experts.hidden, qw, qz, sc, expert_ids, post_pad,
                                     experts.zero_offset, "silu_pack")
    else:
        # small grids split K across workgroups: reduce first, then the separate SiluAndMul + pack
        # (columns de-interleaved back to [gate | up]; rows of skipped m-tiles are never consumed)
        h = ops.wna16_gemm_grouped(packed, m_pad, experts.hidden, qw, qz, sc, expert_ids, post_pad,
                                   experts.zero_offset, "out")
        h = h.view(m_pad, n13 // 2, 2).transpose(1, 2).reshape(m_pad, n13).contiguous()
        act = ops.silu_and_mul_pack(h)
    qw, qz, sc = experts.w2
    slabs, _ = ops.wna16_gemm_grouped(act, m_pad, experts.inter, qw, qz, sc, expert_ids, post_pad,
                                      experts.zero_offset, "slabs")
    if defer_combine:
        return DeferredCombine(slabs, inv, topk_weights.contiguous(), hidden_states.dtype)
    return ops.moe_combine(slabs, inv, topk_weights.contiguous(), hidden_states.dtype)


# --------------------------------------------------------------------------------------------------
# the layer + quant-method seam (modeling/layers/fused_moe/layer.py:22-35, 146-300)
# --------------------------------------------------------------------------------------------------
class FusedMoEMethodBase(QuantizeMethodBase):
    """What a FusedMoE layer delegates to (layer.py:22-35): same two entry points as the reference."""

    @abc.abstractmethod
    def create_weights(self, layer: nn.Module, num_experts: int, hidden_size: int, intermediate_size: int,
                       params_dtype: torch.dtype, **extra_weight_attrs):
        raise NotImplementedError

    @abc.abstractmethod
    def apply(self, layer: nn.Module, x: torch.Tensor, router_logits: torch.Tensor, top_k: int,
              renormalize: bool, use_grouped_topk: bool = False, topk_group: Optional[int] = None,
              num_expert_group: Optional[int] = None,
              custom_routing_function: Optional[Callable] = None) -> torch.Tensor:
        """The keyword set FusedMoE.forward passes (layer.py:437-446)."""
        raise NotImplementedError


class Wna16MoEMethod(FusedMoEMethodBase):
    """GPTQ / AWQ int4 experts through the grouped CDNA4 GEMM -- the FusedMoEMethodBase the reference
    lacks for these formats (it runs ``mixtral_quant.py``'s per-expert loop).  Parameters are the
    checkpoint tensors stacked over experts, [w1 | w3] merged along N:

        gptq   w13_qweight [E, H/8, 2I]   w13_qzeros [E, H/g, 2I/8]   w13_scales [E, H/g, 2I]
               w2_qweight  [E, I/8, H]    w2_qzeros  [E, I/g, H/8]    w2_scales  [E, I/g, H]
        awq    w13_qweight [E, H, 2I/8]   ... (N packed)              w2_qweight [E, I, H/8]

    with I the per-rank intermediate size (w1/w3 column-parallel, w2 row-parallel: every rank streams
    1/TP of every active expert -- balanced at decode, unlike expert-parallel placement).  After loading
    they are re-laid once into ``Wna16Experts`` (K-packed, gate/up interleaved) and dropped."""

    def __init__(self, layout: str, group_size: int, desc_act: bool = False):
        if layout not in ("gptq", "awq"):
            raise ValueError(f"unknown int4 expert layout {layout}")
        if desc_act:
            raise NotImplementedError("act-order (desc_act) GPTQ experts are not supported by the grouped kernel")
        if group_size != -1 and (group_size <= 0 or group_size % 128 or (group_size // 128) & (group_size // 128 - 1)):
            # (refused here, at construction, not by the kernel's plan at the first forward)
            raise NotImplementedError(f"int4 experts with group size {group_size}: the grouped kernel serves 128 x 2^n")
        self.layout, self.group_size = layout, group_size

    def create_weights(self, layer: nn.Module, num_experts: int, hidden_size: int, intermediate_size: int,
                       params_dtype: torch.dtype, **extra_weight_attrs):
        loader = extra_weight_attrs.get("weight_loader")
        g = self.group_size if self.group_size != -1 else None
        e, h, i = num_experts, hidden_size, intermediate_size
        if g is None or h % g or i % g or i % 8 or h % 8:
            raise ValueError(f"group size {self.group_size} must divide hidden {h} and the per-rank intermediate "
                             f"size {i} (too large a tensor-parallel size?)")

        def reg(name, shape, dtype, in_dim, out_dim, packed_dim=None):
            layer.register_parameter(name, _param(torch.empty(shape, dtype=dtype), input_dim=in_dim,
                                                  output_dim=out_dim, packed_dim=packed_dim,
                                                  pack_factor=8 if packed_dim is not None else None,
                                                  weight_loader=loader))
        # dims are those of ONE expert's [K.., N..] tensor (the expert axis is indexed away first)
        if self.layout == "gptq":
            reg("w13_qweight", (e, h // 8, 2 * i), torch.int32, 0, 1, 0)
            reg("w2_qweight", (e, i // 8, h), torch.int32, 0, 1, 0)
            reg("w13_g_idx", (e, h), torch.int32, 0, None)
            reg("w2_g_idx", (e, i), torch.int32, 0, None)
        else:
            reg("w13_qweight", (e, h, 2 * i // 8), torch.int32, 0, 1, 1)
            reg("w2_qweight", (e, i, h // 8), torch.int32, 0, 1, 1)
        reg("w13_qzeros", (e, h // g, 2 * i // 8), torch.int32, 0, 1, 1)
        reg("w13_scales", (e, h // g, 2 * i), params_dtype, 0, 1)
        reg("w2_qzeros", (e, i // g, h // 8), torch.int32, 0, 1, 1)
        reg("w2_scales", (e, i // g, h), params_dtype, 0, 1)
        layer.experts_packed = None

    def process_weights_after_loading(self, layer: nn.Module) -> None:
        e = layer.w13_qweight.shape[0]
        if self.layout == "gptq":
            for name in ("w13_g_idx", "w2_g_idx"):      # desc_act = False: rows must be in group order
                gi = getattr(layer, name).data
                k = gi.shape[1]
                tp_rank = getattr(layer, "tp_rank", None)       # (the reference's FusedMoE keeps only tp_size)
                if tp_rank is None:
                    from .distributed import get_tensor_model_parallel_rank
                    tp_rank = get_tensor_model_parallel_rank() if getattr(layer, "tp_size", 1) > 1 else 0
                k0 = tp_rank * k if name == "w2_g_idx" else 0
                want = ((torch.arange(k, device=gi.device) + k0) // self.group_size).to(gi.dtype)
                if not torch.equal(gi, want.expand_as(gi)):
                    raise ValueError(f"{name}: act-order g_idx found in a checkpoint declared desc_act=false")
        sets13 = [(layer.w13_qweight.data[x], layer.w13_qzeros.data[x], layer.w13_scales.data[x]) for x in range(e)]
        sets2 = [(layer.w2_qweight.data[x], layer.w2_qzeros.data[x], layer.w2_scales.data[x]) for x in range(e)]
        layer.experts_packed = Wna16Experts(sets13, sets2, zero_offset=1 if self.layout == "gptq" else 0,
                                            layout=self.layout)
        for name in ("w13_qweight", "w13_qzeros", "w13_scales", "w2_qweight", "w2_qzeros", "w2_scales",
                     "w13_g_idx", "w2_g_idx"):
            if hasattr(layer, name):
                delattr(layer, name)

    def apply(self, layer: nn.Module, x: torch.Tensor, router_logits: torch.Tensor, top_k: int,
              renormalize: bool, use_grouped_topk: bool = False, topk_group: Optional[int] = None,
              num_expert_group: Optional[int] = None, custom_routing_function: Optional[Callable] = None,
              defer_combine: bool = False) -> torch.Tensor:
        if use_grouped_topk:
            raise NotImplementedError("grouped top-k routing (DeepSeek-V2) is outside the hot path")
        if layer.experts_packed is None:
            raise RuntimeError("FusedMoE: process_weights_after_loading has not run")
        if custom_routing_function is None and not getattr(layer, "record_routing", False):
            # (routing, alignment and the packed gather are one launch where csrc/moe.hip serves the shape: fused_wna16_moe)
            return fused_wna16_moe(x, layer.experts_packed, router_logits, top_k, renormalize, defer_combine=defer_combine)
        topk_weights, topk_ids, sorted_ids, expert_ids, post_pad, inv = route_and_align(
            x, router_logits, top_k, renormalize, layer.experts_packed.num_experts, want_inverse=True,
            custom_routing_function=custom_routing_function)
        if getattr(layer, "record_routing", False):     # measurement aid: which experts a step touched
            layer.last_topk_ids = topk_ids
        return fused_wna16_moe(x, layer.experts_packed, router_logits, top_k, renormalize,
                               topk_weights=topk_weights, topk_ids=topk_ids,
                               aligned=(sorted_ids, expert_ids, post_pad, inv), defer_combine=defer_combine)


class CompressedTensorsMoEMethod(Wna16MoEMethod):
    """compressed-tensors ``pack-quantized`` int4 experts (llm-compressor W4A16 Mixtral checkpoints): the role of
    quantization/compressed_tensors/compressed_tensors_moe.py:23-286, which repacks for ``fused_marlin_moe`` (CUDA only).
    Same parameters and names as the reference (:48-157) -- the loader transposes the on-disk [N, K / 8] / [N, G]
    tensors (``is_transposed``, fused_moe/layer.py:324-331), so that

        w13_weight_packed [E, H/8, 2I] int32     w13_weight_scale [E, H/g, 2I]     w13_weight_shape [E, 2]
        w2_weight_packed  [E, I/8, H]  int32     w2_weight_scale  [E, I/g, H]      w2_weight_shape  [E, 2]

    -- and a transposed ``weight_packed`` IS AutoGPTQ's qweight (nibble i of word r = element k = 8 r + i, stored q + 8:
    compressed_tensors_wNA16.py:97-135, uint4b8).  Symmetric weights have the zero point 8 in every group, i.e. 7 in
    GPTQ's stored-minus-one convention: after loading the tensors go through ``Wna16Experts`` unchanged and the grouped
    CDNA4 GEMM serves them like GPTQ experts.  ``channel`` strategy: one scale row per expert matrix, expanded to groups
    of 128 (the grouped kernel's group sizes are 128 x 2^n and K = 14336 is not one).  8-bit and act-order are refused."""

    def __init__(self, quant_config):
        scheme = quant_config.target_scheme_map.get("Linear")
        w = scheme.get("weights") if scheme else None
        if w is None:
            raise ValueError("compressed-tensors experts: no `Linear` target with a weights block in the config")
        if not w.symmetric:
            raise ValueError("Only symmetric quantization is supported for MoE")             # (:37-38)
        if quant_config.quant_format != "pack-quantized" or w.num_bits not in (4, 8) or w.type != "int":
            raise ValueError("For Fused MoE layers, only pack-quantized is supported for the following bits: [4, 8]")
        if w.num_bits != 4:
            raise NotImplementedError("pack-quantized 8-bit experts: only 4-bit is built for MI355X")
        if w.actorder is not None:
            raise NotImplementedError("act-order compressed-tensors experts are not supported by the grouped kernel")
        if w.strategy not in ("group", "channel"):
            raise ValueError(f"compressed-tensors experts: unsupported weight strategy {w.strategy}")
        self.strategy = w.strategy
        self.ckpt_group_size = w.group_size if w.strategy == "group" else -1
        super().__init__("gptq", w.group_size if w.strategy == "group" else 128, False)

    def create_weights(self, layer: nn.Module, num_experts: int, hidden_size: int, intermediate_size: int,
                       params_dtype: torch.dtype, **extra_weight_attrs):
        loader = extra_weight_attrs.get("weight_loader")
        e, h, i = num_experts, hidden_size, intermediate_size
        g = self.ckpt_group_size
        if i % 8 or h % 8 or (g != -1 and (g <= 0 or h % g or i % g)) or (g == -1 and (h % 128 or i % 128)):
            raise ValueError(f"group size {g} must divide hidden {h} and the per-rank intermediate size {i} "
                             f"(too large a tensor-parallel size?)")
        g13, g2 = (1, 1) if g == -1 else (h // g, i // g)

        def reg(name, shape, dtype, in_dim, out_dim, packed_dim=None):
            layer.register_parameter(name, _param(torch.empty(shape, dtype=dtype), input_dim=in_dim,
                                                  output_dim=out_dim, packed_dim=packed_dim,
                                                  pack_factor=8 if packed_dim is not None else None,
                                                  weight_loader=loader, is_transposed=True,
                                                  quant_method=self.strategy))
        reg("w13_weight_packed", (e, h // 8, 2 * i), torch.int32, 0, 1, 0)
        reg("w2_weight_packed", (e, i // 8, h), torch.int32, 0, 1, 0)
        # channel strategy: ONE scale row; w2's is not cut by tensor parallelism (layer.py:252-265)
        reg("w13_weight_scale", (e, g13, 2 * i), params_dtype, 0 if g != -1 else None, 1)
        reg("w2_weight_scale", (e, g2, h), params_dtype, 0 if g != -1 else None, 1)
        for name in ("w13_weight_shape", "w2_weight_shape"):
            layer.register_parameter(name, _param(torch.zeros((e, 2), dtype=torch.int64), weight_loader=loader))
        layer.experts_packed = None

    def process_weights_after_loading(self, layer: nn.Module) -> None:
        e = layer.w13_weight_packed.shape[0]

        def tensors(packed, scale):
            k8, n = packed.shape
            groups = k8 * 8 // self.group_size
            if scale.shape[0] != groups:                                   # channel strategy: one row -> groups of 128
                scale = scale.expand(groups, n)
            zeros = torch.full((groups, n // 8), 0x77777777, dtype=torch.int32, device=packed.device)
            return packed, zeros, scale.contiguous()
        sets13 = [tensors(layer.w13_weight_packed.data[x], layer.w13_weight_scale.data[x]) for x in range(e)]
        sets2 = [tensors(layer.w2_weight_packed.data[x], layer.w2_weight_scale.data[x]) for x in range(e)]
        layer.experts_packed = Wna16Experts(sets13, sets2, zero_offset=1, layout="gptq")
        for name in ("w13_weight_packed", "w13_weight_scale", "w13_weight_shape", "w2_weight_packed", "w2_weight_scale",
                     "w2_weight_shape"):
            delattr(layer, name)


def fused_fp8_moe(hidden_states: torch.Tensor, w13: torch.Tensor, w2: torch.Tensor, w13_scale: torch.Tensor,
                  w2_scale: torch.Tensor, topk_weights: torch.Tensor, topk_ids: torch.Tensor,
                  a1_scale: Optional[torch.Tensor] = None, a2_scale: Optional[torch.Tensor] = None,
                  aligned=None) -> torch.Tensor:
    """``fused_experts(..., use_fp8_w8a8=True)`` (fused_moe.py:566-690) on the grouped FP8 kernel (csrc/fp8_moe.hip).
    w13 [E, 2I, H] / w2 [E, H, I] e4m3, one weight scale per expert, per-tensor activation scales (static, or dynamic
    over the whole tensor: ``scaled_fp8_quant``), the reference's intermediate layouts and roundings:

        A1_q, s1 = scaled_fp8_quant(hidden)            cache1[slot] = T(acc * s1 * w13_scale[e])      slot = token * k + j
        cache2   = silu_and_mul(cache1)                A2_q, s2 = scaled_fp8_quant(cache2)
        cache3[slot] = T((acc * w_routed[slot]) * s2 * w2_scale[e])                out = sum_j cache3[token, j]"""
    m, h = hidden_states.shape
    e, n13, _ = w13.shape
    k = topk_ids.shape[1]
    dev, dt = hidden_states.device, hidden_states.dtype
    if aligned is not None:
        sorted_ids, expert_ids, post_pad = aligned
    else:
        sorted_ids, expert_ids, post_pad = moe_align_block_size(topk_ids, MOE_BLOCK_M, e)
    xq, s1 = ops.scaled_fp8_quant(hidden_states, a1_scale)
    cache1 = torch.empty((m * k, n13), dtype=dt, device=dev)
    ops.fp8_moe_gemm(xq, w13, s1, w13_scale, None, sorted_ids, expert_ids, post_pad, cache1, k)
    cache2 = torch.empty((m * k, n13 // 2), dtype=dt, device=dev)
    ops.silu_and_mul(cache2, cache1)
    aq, s2 = ops.scaled_fp8_quant(cache2, a2_scale)
    cache3 = torch.empty((m * k, h), dtype=dt, device=dev)
    ops.fp8_moe_gemm(aq, w2, s2, w2_scale, topk_weights.reshape(-1).contiguous(), sorted_ids, expert_ids, post_pad,
                     cache3, 1)
    return cache3.view(m, k, h).sum(dim=1)


class Fp8MoEMethod(FusedMoEMethodBase):
    """quantization/fp8.py:279-503 (Fp8MoEMethod): FP8 checkpoints with one weight scale per expert matrix and static or
    dynamic per-tensor activation scales, or 16-bit checkpoints quantised at load.  Same parameters, same
    post-processing (a single w13 scale per expert: max of the w1 / w3 scales, both halves requantised to it; static input
    scales reduced to their maximum over the experts).  gfx950 computes in OCP e4m3fn: no fnuz renormalisation."""

    def __init__(self, quant_config):
        self.quant_config = quant_config

    def create_weights(self, layer: nn.Module, num_experts: int, hidden_size: int, intermediate_size: int,
                       params_dtype: torch.dtype, **extra_weight_attrs):
        loader = extra_weight_attrs.get("weight_loader")
        cfg = self.quant_config
        layer.orig_dtype = params_dtype           # (our FusedMoE sets it too; the reference's layer does not)
        wdt = torch.float8_e4m3fn if cfg.is_checkpoint_fp8_serialized else params_dtype
        e, h, i = num_experts, hidden_size, intermediate_size
        if h % 128 or i % 128:
            raise ValueError(f"FP8 experts need hidden ({h}) and per-rank intermediate ({i}) sizes that are multiples of 128")
        layer.register_parameter("w13_weight", _param(torch.empty(e, 2 * i, h, dtype=wdt), input_dim=1, output_dim=0,
                                                      weight_loader=loader))
        layer.register_parameter("w2_weight", _param(torch.empty(e, h, i, dtype=wdt), input_dim=1, output_dim=0,
                                                     weight_loader=loader))
        # two scales for w1 and w3, combined after loading (fp8.py:319-331)
        ser = cfg.is_checkpoint_fp8_serialized
        # (quant_method = "tensor": how the reference's FusedMoE.weight_loader recognises per-tensor scales, fp8.py:333-336,
        #  fused_moe/layer.py:340-361)
        layer.register_parameter("w13_weight_scale", _param(torch.ones(e, 2, dtype=torch.float32),
                                                            weight_loader=loader if ser else None,
                                                            quant_method="tensor" if ser else None))
        layer.register_parameter("w2_weight_scale", _param(torch.ones(e, dtype=torch.float32),
                                                           weight_loader=loader if ser else None,
                                                           quant_method="tensor" if ser else None))
        if cfg.activation_scheme == "static":
            if not ser:
                raise ValueError("Found static activation scheme for checkpoint that was not serialized fp8.")
            layer.register_parameter("w13_input_scale", _param(torch.ones(e, dtype=torch.float32), weight_loader=loader))
            layer.register_parameter("w2_input_scale", _param(torch.ones(e, dtype=torch.float32), weight_loader=loader))
        else:
            layer.w13_input_scale = None
            layer.w2_input_scale = None

    def process_weights_after_loading(self, layer: nn.Module) -> None:
        cfg = self.quant_config
        e = layer.w13_weight.shape[0]
        if not cfg.is_checkpoint_fp8_serialized:        # 16-bit checkpoint: one scale per merged w13 / per w2 (fp8.py:370-396)
            w13 = torch.empty_like(layer.w13_weight.data, dtype=torch.float8_e4m3fn)
            w2 = torch.empty_like(layer.w2_weight.data, dtype=torch.float8_e4m3fn)
            s13 = torch.ones(e, dtype=torch.float32, device=w13.device)
            s2 = torch.ones(e, dtype=torch.float32, device=w13.device)
            for x in range(e):
                q, sc = ops.scaled_fp8_quant(layer.w13_weight.data[x])
                w13[x].copy_(q)
                s13[x] = sc.reshape(())
                q, sc = ops.scaled_fp8_quant(layer.w2_weight.data[x])
                w2[x].copy_(q)
                s2[x] = sc.reshape(())
            layer.w13_weight = nn.Parameter(w13, requires_grad=False)
            layer.w2_weight = nn.Parameter(w2, requires_grad=False)
            layer.w13_weight_scale = nn.Parameter(s13, requires_grad=False)
            layer.w2_weight_scale = nn.Parameter(s2, requires_grad=False)
            return
        if cfg.activation_scheme == "static":            # a single activation scale: the maximum (fp8.py:404-419)
            if layer.w13_input_scale is None or layer.w2_input_scale is None:
                raise ValueError("QuantConfig has static quantization, but found activation scales are None.")
            layer.w13_input_scale = nn.Parameter(layer.w13_input_scale.data.max().reshape(1), requires_grad=False)
            layer.w2_input_scale = nn.Parameter(layer.w2_input_scale.data.max().reshape(1), requires_grad=False)
        # a single weight scale for w13 per expert: take the max, dequantise each half with its own scale and requantise
        # (fp8.py:447-465)
        shard = layer.intermediate_size_per_partition
        mx = layer.w13_weight_scale.data.max(dim=1).values
        w13 = layer.w13_weight.data
        for x in range(e):
            for sh in range(2):
                rows = w13[x, sh * shard:(sh + 1) * shard, :]
                # per_tensor_dequantize (w8a8_utils.py:23-28) widens to FLOAT16 whatever the model dtype and multiplies
                # there; the requantisation reads that f16 tensor (rounding through bf16 changed some e4m3 words)
                dq = rows.to(torch.float16) * layer.w13_weight_scale.data[x, sh]
                q, _ = ops.scaled_fp8_quant(dq, mx[x].reshape(1))
                rows.copy_(q)
        layer.w13_weight_scale = nn.Parameter(mx.contiguous(), requires_grad=False)
        layer.w13_weight = nn.Parameter(w13, requires_grad=False)
        layer.w2_weight = nn.Parameter(layer.w2_weight.data, requires_grad=False)
        layer.w2_weight_scale = nn.Parameter(layer.w2_weight_scale.data, requires_grad=False)

    def apply(self, layer: nn.Module, x: torch.Tensor, router_logits: torch.Tensor, top_k: int,
              renormalize: bool, use_grouped_topk: bool = False, topk_group: Optional[int] = None,
              num_expert_group: Optional[int] = None,
              custom_routing_function: Optional[Callable] = None) -> torch.Tensor:
        if use_grouped_topk:
            raise NotImplementedError("grouped top-k routing (DeepSeek-V2) is outside the hot path")
        topk_weights, topk_ids, sorted_ids, expert_ids, post_pad, _ = route_and_align(
            x, router_logits, top_k, renormalize, layer.w13_weight.shape[0],
            custom_routing_function=custom_routing_function)
        if getattr(layer, "record_routing", False):
            layer.last_topk_ids = topk_ids
        return fused_fp8_moe(x, layer.w13_weight, layer.w2_weight, layer.w13_weight_scale, layer.w2_weight_scale,
                             topk_weights, topk_ids, layer.w13_input_scale, layer.w2_input_scale,
                             aligned=(sorted_ids, expert_ids, post_pad))


class FusedMoE(nn.Module):
    """modeling/layers/fused_moe/layer.py:146-300 for quantised experts: owns the stacked expert
    parameters, shards them over the TP group (intermediate dimension) and optionally all-reduces."""

    SHARDS = ("w1", "w2", "w3")

    def __init__(self, num_experts: int, top_k: int, hidden_size: int, intermediate_size: int,
                 params_dtype: torch.dtype = torch.float16, reduce_results: bool = False, renormalize: bool = True,
                 quant_config=None, tp_size: Optional[int] = None, prefix: str = ""):
        super().__init__()
        from .distributed import get_tensor_model_parallel_rank, get_tensor_model_parallel_world_size
        self.tp_size = tp_size if tp_size is not None else get_tensor_model_parallel_world_size()
        self.tp_rank = get_tensor_model_parallel_rank() if self.tp_size > 1 else 0
        if intermediate_size % self.tp_size != 0:
            raise ValueError(f"intermediate size {intermediate_size} is not divisible by tensor parallel size "
                             f"{self.tp_size}")
        self.num_experts, self.top_k = num_experts, top_k
        self.hidden_size = hidden_size
        self.intermediate_size_per_partition = intermediate_size // self.tp_size
        self.reduce_results, self.renormalize = reduce_results, renormalize
        if quant_config is None:
            raise NotImplementedError("unquantised experts are outside the hot path (SURVEY 8f row 2: quantised MoE)")
        self.quant_method = quant_config.get_quant_method(self, prefix)
        if not isinstance(self.quant_method, FusedMoEMethodBase):
            raise NotImplementedError(f"{type(quant_config).__name__} has no FusedMoE method")
        self.orig_dtype = params_dtype
        self.quant_method.create_weights(layer=self, num_experts=num_experts, hidden_size=hidden_size,
                                         intermediate_size=self.intermediate_size_per_partition,
                                         params_dtype=params_dtype, weight_loader=self.weight_loader)

    def weight_loader(self, param: nn.Parameter, loaded_weight: torch.Tensor, weight_name: str, shard_id: str,
                      expert_id: int) -> None:
        """One checkpoint tensor of one expert -> its slot.  w1 / w3 are cut along their output
        dimension (this rank's slice lands in the first / second half of the merged w13 parameter),
        w2 along its input dimension; packing follows the parameter's ``packed_dim``."""
        if shard_id not in self.SHARDS:
            raise ValueError(f"shard_id must be ['w1','w2','w3'] but got {shard_id}.")
        if not 0 <= expert_id < self.num_experts:
            raise ValueError(f"{weight_name}: expert {expert_id} out of range")
        if getattr(param, "is_transposed", False):     # compressed-tensors experts: [N, ...] on disk (layer.py:324-331)
            loaded_weight = loaded_weight.t().contiguous()
        elif "weight_shape" in weight_name:            # (layer.py:365-369)
            param.data[expert_id] = loaded_weight.reshape(-1).to(param.data.dtype)
            return
        if "weight_scale" in weight_name and not getattr(param, "is_transposed", False):
            # per-tensor scales (layer.py:219-232): w1 / w3 kept apart
            if param.data.dim() == 2:
                if shard_id == "w2":
                    raise ValueError(f"{weight_name}: a [E, 2] scale parameter belongs to w1 / w3")
                param.data[expert_id][0 if shard_id == "w1" else 1] = loaded_weight.reshape(())
            else:
                param.data[expert_id] = loaded_weight.reshape(())
            return
        if "input_scale" in weight_name:               # (layer.py:299-304; w1 and w3 of a layer must agree)
            # (checked whichever of w1 / w3 arrives second, as the reference does: 1 is the "nothing loaded yet" value)
            if shard_id in ("w1", "w3") and param.data[expert_id] != 1 and \
                    (param.data[expert_id] - loaded_weight.reshape(())).abs() > 1e-5:
                raise ValueError("input_scales of w1 and w3 of a layer must be equal. But got "
                                 f"{param.data[expert_id]} vs. {loaded_weight}")
            param.data[expert_id] = loaded_weight.reshape(())
            return
        data = param.data[expert_id]
        in_dim, out_dim = getattr(param, "input_dim", None), getattr(param, "output_dim", None)
        if shard_id == "w2":
            if in_dim is not None and self.tp_size > 1:
                size = data.shape[in_dim]
                loaded_weight = loaded_weight.narrow(in_dim, self.tp_rank * size, size)
            dst = data
        elif out_dim is None:                      # replicated per-expert metadata (g_idx of w1 / w3)
            dst = data
        else:
            half = data.shape[out_dim] // 2
            loaded_weight = loaded_weight.narrow(out_dim, self.tp_rank * half, half)
            dst = data.narrow(out_dim, 0 if shard_id == "w1" else half, half)
        if dst.shape != loaded_weight.shape:
            raise ValueError(f"{weight_name}: checkpoint tensor {tuple(loaded_weight.shape)} does not fit "
                             f"parameter slice {tuple(dst.shape)}")
        dst.copy_(loaded_weight)

    @classmethod
    def make_expert_params_mapping(cls, ckpt_gate_proj_name: str, ckpt_down_proj_name: str,
                                   ckpt_up_proj_name: str, num_experts: int) -> List[Tuple[str, str, int, str]]:
        """(param_name_prefix, checkpoint_name_prefix, expert_id, shard_id) rows (layer.py:449-468)."""
        rows = []
        for expert in range(num_experts):
            for shard, ckpt in (("w1", ckpt_gate_proj_name), ("w2", ckpt_down_proj_name),
                                ("w3", ckpt_up_proj_name)):
                fused = "experts.w13_" if shard in ("w1", "w3") else "experts.w2_"
                rows.append((fused, f"experts.{expert}.{ckpt}.", expert, shard))
        return rows

    def forward(self, hidden_states: torch.Tensor, router_logits: torch.Tensor, defer_combine: bool = False,
                defer_all_reduce: bool = False):
        """defer_combine: return a DeferredCombine for the next norm launch to consume (int4 experts on one rank only).
        defer_all_reduce (TP > 1): return a distributed.DeferredAllReduce where the fused all-reduce + norm launch serves
        the size -- the next norm runs the sum."""
        if defer_combine and isinstance(self.quant_method, Wna16MoEMethod) and not (self.reduce_results and self.tp_size > 1):
            return self.quant_method.apply(self, hidden_states, router_logits, self.top_k, self.renormalize,
                                           defer_combine=True)
        out = self.quant_method.apply(self, hidden_states, router_logits, self.top_k, self.renormalize)
        if self.reduce_results and self.tp_size > 1:
            from .distributed import defer_all_reduce as _defer, tensor_model_parallel_all_reduce
            if defer_all_reduce and out.dim() == 2:
                dar = _defer(out if out.is_contiguous() else out.contiguous())
                if dar is not None:
                    return dar
            out = tensor_model_parallel_all_reduce(out)
        return out
