"""PagedAttention helper -- mirror of aphrodite/attention/ops/paged_attn.py:33-253
(cache shape/split :40-62, write :65-84, decode dispatch :87-190) calling the
MI355X ops."""
from typing import List, Optional, Tuple

import torch

from .. import _custom_ops as ops

_PARTITION_SIZE = 512  # paged_attn.py:13


class PagedAttention:
    @staticmethod
    def get_supported_head_sizes() -> List[int]:
        return [64, 80, 96, 112, 128, 192, 256]

    @staticmethod
    def get_kv_cache_shape(num_blocks: int, block_size: int, num_kv_heads: int,
                           head_size: int) -> Tuple[int, ...]:
        return (2, num_blocks, block_size * num_kv_heads * head_size)

    @staticmethod
    def split_kv_cache(kv_cache: torch.Tensor, num_kv_heads: int,
                       head_size: int) -> Tuple[torch.Tensor, torch.Tensor]:
        x = 16 // kv_cache.element_size()
        num_blocks = kv_cache.shape[1]
        key_cache = kv_cache[0].view(num_blocks, num_kv_heads, head_size // x, -1, x)
        value_cache = kv_cache[1].view(num_blocks, num_kv_heads, head_size, -1)
        return key_cache, value_cache

    @staticmethod
    def write_to_paged_cache(key, value, key_cache, value_cache, slot_mapping,
                             kv_cache_dtype: str, k_scale: float, v_scale: float) -> None:
        ops.reshape_and_cache(key, value, key_cache, value_cache,
                              slot_mapping.flatten(), kv_cache_dtype, k_scale, v_scale)

    @staticmethod
    def forward_decode(query, key_cache, value_cache, block_tables, seq_lens,
                       max_seq_len: int, kv_cache_dtype: str, num_kv_heads: int,
                       scale: float, alibi_slopes: Optional[torch.Tensor],
                       k_scale: float, v_scale: float, tp_rank: int = 0,
                       blocksparse_local_blocks: int = 0,
                       blocksparse_vert_stride: int = 0,
                       blocksparse_block_size: int = 64,
                       blocksparse_head_sliding_step: int = 0) -> torch.Tensor:
        output = torch.empty_like(query)
        block_size = value_cache.shape[3]
        num_seqs, num_heads, head_size = query.shape
        max_num_partitions = (max_seq_len + _PARTITION_SIZE - 1) // _PARTITION_SIZE
        # v1/v2 heuristic of the reference (paged_attn.py:121-128)
        use_v1 = (max_seq_len <= 8192
                  and (max_num_partitions == 1 or num_seqs * num_heads > 512))
        if use_v1:
            ops.paged_attention_v1(output, query, key_cache, value_cache,
                                   num_kv_heads, scale, block_tables, seq_lens,
                                   block_size, max_seq_len, alibi_slopes,
                                   kv_cache_dtype, k_scale, v_scale, tp_rank,
                                   blocksparse_local_blocks, blocksparse_vert_stride,
                                   blocksparse_block_size, blocksparse_head_sliding_step)
        else:
            assert _PARTITION_SIZE % block_size == 0
            tmp_output = torch.empty(
                size=(num_seqs, num_heads, max_num_partitions, head_size),
                dtype=output.dtype, device=output.device)
            exp_sums = torch.empty(size=(num_seqs, num_heads, max_num_partitions),
                                   dtype=torch.float32, device=output.device)
            max_logits = torch.empty_like(exp_sums)
            ops.paged_attention_v2(output, exp_sums, max_logits, tmp_output, query,
                                   key_cache, value_cache, num_kv_heads, scale,
                                   block_tables, seq_lens, block_size, max_seq_len,
                                   alibi_slopes, kv_cache_dtype, k_scale, v_scale,
                                   tp_rank, blocksparse_local_blocks,
                                   blocksparse_vert_stride, blocksparse_block_size,
                                   blocksparse_head_sliding_step)
        return output

    @staticmethod
    def forward_prefix(query, key, value, kv_cache_dtype: str, key_cache, value_cache,
                       block_tables, query_start_loc, seq_lens_tensor, context_lens,
                       max_query_len: int, alibi_slopes, sliding_window, k_scale: float,
                       v_scale: float, max_seq_len: Optional[int] = None,
                       total_kv_tokens: Optional[int] = None) -> torch.Tensor:
        """ops/paged_attn.py:192-231.  ``max_seq_len`` / ``total_kv_tokens``: host-side max / sum of the prefill
        sequences' lengths (the metadata holds them as a Python list): with them the long-prompt path of
        context_attention_fwd needs no device-to-host sync."""
        output = torch.empty_like(query)
        ops.context_attention_fwd(query, key, value, output, kv_cache_dtype, key_cache,
                                  value_cache, block_tables, query_start_loc, seq_lens_tensor,
                                  context_lens, max_query_len, k_scale, v_scale, alibi_slopes,
                                  sliding_window, max_seq_len=max_seq_len, total_kv_tokens=total_kv_tokens)
        return output

    @staticmethod
    def swap_blocks(src_kv_cache: torch.Tensor, dst_kv_cache: torch.Tensor,
                    src_to_dst: torch.Tensor) -> None:
        """ops/paged_attn.py:233-244: kv cache [2, NB, ...]; keys then values."""
        ops.swap_blocks(src_kv_cache[0], dst_kv_cache[0], src_to_dst)
        ops.swap_blocks(src_kv_cache[1], dst_kv_cache[1], src_to_dst)

    @staticmethod
    def copy_blocks(kv_caches: List[torch.Tensor], src_to_dists: torch.Tensor) -> None:
        """ops/paged_attn.py:246-253."""
        key_caches = [kv_cache[0] for kv_cache in kv_caches]
        value_caches = [kv_cache[1] for kv_cache in kv_caches]
        ops.copy_blocks(key_caches, value_caches, src_to_dists)
