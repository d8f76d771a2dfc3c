"""MI355X attention backend -- mirrors the plugin types of
aphrodite/attention/backends/abstract.py:21-231 and the behaviour of
ROCmFlashAttention{Backend,Metadata,Impl}
(aphrodite/attention/backends/rocm_flash_attn.py:26-595): cache write ->
prefill (varlen causal attention) -> decode (paged attention over the cache)."""
from dataclasses import dataclass
from typing import List, Optional, Tuple, Type

import torch

from .. import _custom_ops as ops
from .paged_attn import PagedAttention

_PARTITION_SIZE_ROCM = 512  # rocm_flash_attn.py:23


@dataclass
class MI355XAttentionMetadata:
    """Fields of ROCmFlashAttentionMetadata (rocm_flash_attn.py:74-122)."""
    num_prefills: int
    num_prefill_tokens: int
    num_decode_tokens: int
    slot_mapping: torch.Tensor
    seq_lens: Optional[List[int]]
    seq_lens_tensor: Optional[torch.Tensor]
    max_query_len: Optional[int]
    max_prefill_seq_len: int
    max_decode_seq_len: int
    query_start_loc: Optional[torch.Tensor]
    seq_start_loc: Optional[torch.Tensor]
    context_lens_tensor: Optional[torch.Tensor]
    block_tables: Optional[torch.Tensor]
    use_cuda_graph: bool = False

    @property
    def prefill_metadata(self) -> Optional["MI355XAttentionMetadata"]:
        if self.num_prefills == 0:
            return None
        return MI355XAttentionMetadata(
            num_prefills=self.num_prefills,
            num_prefill_tokens=self.num_prefill_tokens, num_decode_tokens=0,
            slot_mapping=self.slot_mapping[:self.num_prefill_tokens],
            seq_lens=self.seq_lens[:self.num_prefills] if self.seq_lens else None,
            seq_lens_tensor=self.seq_lens_tensor[:self.num_prefills],
            max_query_len=self.max_query_len,
            max_prefill_seq_len=self.max_prefill_seq_len, max_decode_seq_len=0,
            query_start_loc=self.query_start_loc[:self.num_prefills + 1],
            seq_start_loc=(self.seq_start_loc[:self.num_prefills + 1]
                           if self.seq_start_loc is not None else None),
            context_lens_tensor=(self.context_lens_tensor[:self.num_prefills]
                                 if self.context_lens_tensor is not None else None),
            block_tables=(self.block_tables[:self.num_prefills]
                          if self.block_tables is not None else None),
            use_cuda_graph=False)

    @property
    def decode_metadata(self) -> Optional["MI355XAttentionMetadata"]:
        if self.num_decode_tokens == 0:
            return None
        return MI355XAttentionMetadata(
            num_prefills=0, num_prefill_tokens=0,
            num_decode_tokens=self.num_decode_tokens,
            slot_mapping=self.slot_mapping[self.num_prefill_tokens:],
            seq_lens=None,
            seq_lens_tensor=self.seq_lens_tensor[self.num_prefills:],
            max_query_len=None, max_prefill_seq_len=0,
            max_decode_seq_len=self.max_decode_seq_len,
            query_start_loc=None, seq_start_loc=None, context_lens_tensor=None,
            block_tables=self.block_tables[self.num_prefills:],
            use_cuda_graph=self.use_cuda_graph)


class MI355XAttentionBackend:
    @staticmethod
    def get_name() -> str:
        return "mi355x-paged-attn"

    @staticmethod
    def get_impl_cls() -> Type["MI355XAttentionImpl"]:
        return MI355XAttentionImpl

    @staticmethod
    def get_metadata_cls() -> Type["MI355XAttentionMetadata"]:
        return MI355XAttentionMetadata

    @classmethod
    def make_metadata(cls, *args, **kwargs) -> "MI355XAttentionMetadata":
        return cls.get_metadata_cls()(*args, **kwargs)

    @staticmethod
    def get_kv_cache_shape(num_blocks: int, block_size: int, num_kv_heads: int,
                           head_size: int) -> Tuple[int, ...]:
        return PagedAttention.get_kv_cache_shape(num_blocks, block_size,
                                                 num_kv_heads, head_size)

    @staticmethod
    def swap_blocks(src_kv_cache, dst_kv_cache, src_to_dst) -> None:
        PagedAttention.swap_blocks(src_kv_cache, dst_kv_cache, src_to_dst)

    @staticmethod
    def copy_blocks(kv_caches, src_to_dists) -> None:
        PagedAttention.copy_blocks(kv_caches, src_to_dists)


class MI355XAttentionImpl:
    """AttentionImpl (abstract.py:181-231): forward(query, key, value, kv_cache,
    attn_metadata, k_scale, v_scale) -> [num_tokens, num_heads * head_size]."""

    def __init__(self, num_heads: int, head_size: int, scale: float,
                 num_kv_heads: int, alibi_slopes: Optional[List[float]] = None,
                 sliding_window: Optional[int] = None, kv_cache_dtype: str = "auto",
                 blocksparse_params=None, logits_soft_cap: Optional[float] = None) -> None:
        if blocksparse_params is not None:
            raise ValueError("MI355X backend does not support blocksparse attention.")
        if logits_soft_cap is not None:
            raise ValueError("MI355X backend does not support attention logits soft capping.")
        if sliding_window is not None:
            raise ValueError("MI355X backend does not support sliding window yet.")
        self.num_heads = num_heads
        self.head_size = head_size
        self.scale = float(scale)
        self.num_kv_heads = num_kv_heads
        self.alibi_slopes = (torch.tensor(alibi_slopes, dtype=torch.float32)
                             if alibi_slopes is not None else None)
        self.kv_cache_dtype = kv_cache_dtype
        self.sliding_window = None
        assert self.num_heads % self.num_kv_heads == 0
        self.num_queries_per_kv = self.num_heads // self.num_kv_heads
        supported = PagedAttention.get_supported_head_sizes()
        if head_size not in supported:
            raise ValueError(f"Head size {head_size} is not supported by "
                             f"PagedAttention. Supported head sizes are: {supported}.")

    def forward(self, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor,
                kv_cache: Optional[torch.Tensor], attn_metadata: MI355XAttentionMetadata,
                k_scale: float = 1.0, v_scale: float = 1.0) -> torch.Tensor:
        num_tokens, hidden_size = query.shape
        query = query.view(-1, self.num_heads, self.head_size)
        key = key.view(-1, self.num_kv_heads, self.head_size)
        value = value.view(-1, self.num_kv_heads, self.head_size)
        if self.alibi_slopes is not None and self.alibi_slopes.device != query.device:
            self.alibi_slopes = self.alibi_slopes.to(query.device)

        key_cache = value_cache = None
        if kv_cache is not None and kv_cache.numel() > 0:
            key_cache, value_cache = PagedAttention.split_kv_cache(
                kv_cache, self.num_kv_heads, self.head_size)
            PagedAttention.write_to_paged_cache(
                key, value, key_cache, value_cache, attn_metadata.slot_mapping,
                self.kv_cache_dtype, k_scale, v_scale)

        num_prefill_tokens = attn_metadata.num_prefill_tokens
        num_decode_tokens = attn_metadata.num_decode_tokens
        output = torch.empty_like(query)
        decode_query = query[num_prefill_tokens:]
        query = query[:num_prefill_tokens]
        key = key[:num_prefill_tokens]
        value = value[:num_prefill_tokens]

        if prefill_meta := attn_metadata.prefill_metadata:
            assert prefill_meta.seq_lens is not None
            has_ctx = (prefill_meta.context_lens_tensor is not None
                       and prefill_meta.block_tables is not None
                       and prefill_meta.block_tables.numel() > 0
                       and bool((prefill_meta.context_lens_tensor > 0).any()))
            if has_ctx:
                # prefix-enabled attention (rocm_flash_attn.py:509-527)
                assert key_cache is not None
                output[:num_prefill_tokens] = PagedAttention.forward_prefix(
                    query, key, value, self.kv_cache_dtype, key_cache, value_cache,
                    prefill_meta.block_tables, prefill_meta.query_start_loc,
                    prefill_meta.seq_lens_tensor, prefill_meta.context_lens_tensor,
                    prefill_meta.max_query_len, self.alibi_slopes, self.sliding_window,
                    k_scale, v_scale)
            else:
                out = ops.flash_attn_varlen(
                    query, key, value, prefill_meta.seq_start_loc,
                    prefill_meta.max_prefill_seq_len, self.scale, causal=True,
                    alibi_slopes=self.alibi_slopes)
                output[:num_prefill_tokens] = out

        if decode_meta := attn_metadata.decode_metadata:
            assert key_cache is not None
            num_seqs = decode_query.shape[0]
            block_size = value_cache.shape[3]
            max_seq_len = decode_meta.max_decode_seq_len
            max_num_partitions = ((max_seq_len + _PARTITION_SIZE_ROCM - 1)
                                  // _PARTITION_SIZE_ROCM)
            tmp_output = torch.empty(
                size=(num_seqs, self.num_heads, max_num_partitions, self.head_size),
                dtype=output.dtype, device=output.device)
            exp_sums = torch.empty(size=(num_seqs, self.num_heads, max_num_partitions),
                                   dtype=torch.float32, device=output.device)
            max_logits = torch.empty_like(exp_sums)
            ops.paged_attention_rocm(
                output[num_prefill_tokens:], exp_sums, max_logits, tmp_output,
                decode_query, key_cache, value_cache, self.num_kv_heads, self.scale,
                decode_meta.block_tables, decode_meta.seq_lens_tensor, block_size,
                max_seq_len, self.alibi_slopes, self.kv_cache_dtype, k_scale, v_scale)
        return output.view(num_tokens, hidden_size)
