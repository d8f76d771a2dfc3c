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
from typing import Optional, Tuple

import torch

from . import _custom_ops as ops

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


class Wna16Experts:
    """Stacked int4 expert weights in the layouts the grouped kernel consumes.

    w13: merged [w1 | w3] (gate | up) per expert, K-packed exllama order [E, K/8, 2I] with the gate/up
    columns INTERLEAVED (``ops.interleave_gate_up``) ; w2: [E, I/8, H].  Built from per-expert GPTQ
    tensor sets (qweight [K/8, N] plain GPTQ order, qzeros [G, N/8] stored zero-1, scales [G, N])."""

    def __init__(self, w13_sets, w2_sets, zero_offset: int = 1):
        empty = torch.empty(0, dtype=torch.int32, device=w13_sets[0][0].device)

        def prep(qw, qz, sc, interleave):
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
                    topk_ids: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The fused_marlin_moe role (fused_moe.py:438-542) for GPTQ/AWQ int4 experts."""
    assert hidden_states.shape[1] == experts.hidden, "Hidden size mismatch"
    assert gating_output.shape[1] == experts.num_experts, "Number of experts mismatch"
    m = hidden_states.shape[0]
    if topk_ids is None:
        topk_weights, topk_ids = fused_topk(hidden_states, gating_output, topk, renormalize)
    e = experts.num_experts
    sorted_ids, expert_ids, post_pad, inv = moe_align_block_size(topk_ids, MOE_BLOCK_M, e, want_inverse=True)
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
    return ops.moe_combine(slabs, inv, topk_weights.contiguous(), hidden_states.dtype)
