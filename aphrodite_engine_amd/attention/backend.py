"""MI355X attention backend -- mirrors the plugin types of
aphrodite/attention/backends/abstract.py:21-231 and the behaviour of
ROCmFlashAttention{Backend,Metadata,Impl}
(aphrodite/attention/backends/rocm_flash_attn.py:26-595): cache write ->
prefill (varlen causal attention) -> decode (paged attention over the cache)."""
from contextlib import contextmanager
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Tuple, Type

import numpy as np
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
    # host-side max(context_lens) of the prefill sequences when the builder knows it (ours does): lets the backend
    # choose the cached-context kernel without a device->host sync per layer.  None: the reference's rule applies
    # (block_tables.numel() > 0, rocm_flash_attn.py:455-459).
    max_context_len: Optional[int] = None

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
            use_cuda_graph=False, max_context_len=self.max_context_len)

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

    def advance_step(self, model_input, sampled_token_ids: Optional[torch.Tensor], block_size: int, num_seqs: int,
                     num_queries: int) -> None:
        """Multi-step decoding (rocm_flash_attn.py:185-229): move a decode-only batch one token ahead IN PLACE -- the host
        list of sequence lengths for the real queries (a graph-padded batch has num_seqs > num_queries), and on the device
        input_tokens / input_positions / seq_lens / slot_mapping through ``advance_step_flashattn``
        (prepare_inputs/advance_step.cu), so that the next step needs no host-side input preparation."""
        if num_seqs != num_queries:
            if num_seqs < num_queries or not self.use_cuda_graph:
                raise ValueError("advance_step: a padded batch (num_seqs > num_queries) is a graph-captured one")
        ok = (self.num_prefills == 0 and self.num_prefill_tokens == 0 and self.num_decode_tokens == num_seqs
              and self.slot_mapping.shape == (num_seqs, ) and self.seq_lens is not None and len(self.seq_lens) == num_seqs
              and self.seq_lens_tensor is not None and self.seq_lens_tensor.shape == (num_seqs, )
              and self.max_query_len == 1 and self.max_prefill_seq_len == 0
              and self.max_decode_seq_len == max(self.seq_lens)
              and self.block_tables is not None and self.block_tables.shape[0] == num_seqs)
        if not ok:
            raise ValueError("advance_step: the metadata is not that of a decode-only batch of num_seqs sequences")
        for i in range(num_queries):
            self.seq_lens[i] += 1
        self.max_decode_seq_len = max(self.seq_lens)
        ops.advance_step_flashattn(num_seqs=num_seqs, num_queries=num_queries, block_size=block_size,
                                   input_tokens=model_input.input_tokens, sampled_token_ids=sampled_token_ids,
                                   input_positions=model_input.input_positions, seq_lens=self.seq_lens_tensor,
                                   slot_mapping=self.slot_mapping, block_tables=self.block_tables)


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

    @staticmethod
    def get_state_cls() -> Type["MI355XAttentionState"]:
        return MI355XAttentionState

    @staticmethod
    def get_builder_cls() -> Type["MI355XAttentionMetadataBuilder"]:
        return MI355XAttentionMetadataBuilder

    @classmethod
    def make_metadata(cls, *args, **kwargs) -> "MI355XAttentionMetadata":
        return cls.get_metadata_cls()(*args, **kwargs)

    @classmethod
    def make_metadata_builder(cls, *args, **kwargs) -> "MI355XAttentionMetadataBuilder":
        return cls.get_builder_cls()(*args, **kwargs)

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
        self.num_heads = num_heads
        self.head_size = head_size
        self.scale = float(scale)
        self.num_kv_heads = num_kv_heads
        self.alibi_slopes = (torch.tensor(alibi_slopes, dtype=torch.float32)
                             if alibi_slopes is not None else None)
        self.kv_cache_dtype = kv_cache_dtype
        # rocm_flash_attn.py:321-322: the window reaches the prompt kernels (flash_attn_varlen_func's window_size, the
        # cached-context kernel's sliding_window); decode sees it through the metadata builder's trimmed block tables and
        # clipped sequence lengths (attention/backends/utils.py:150-185), not through the paged-attention kernel
        self.sliding_window = ((sliding_window, sliding_window) if sliding_window is not None else (-1, -1))
        assert self.num_heads % self.num_kv_heads == 0
        self.num_queries_per_kv = self.num_heads // self.num_kv_heads
        supported = PagedAttention.get_supported_head_sizes()
        if head_size not in supported:
            raise ValueError(f"Head size {head_size} is not supported by "
                             f"PagedAttention. Supported head sizes are: {supported}.")

    def forward(self, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor,
                kv_cache: Optional[torch.Tensor], attn_metadata: MI355XAttentionMetadata,
                k_scale: float = 1.0, v_scale: float = 1.0, attn_type=None) -> torch.Tensor:
        """``attn_type``: the reference's Attention layer passes it by keyword (attention/layer.py:99-106); anything but
        AttentionType.DECODER is refused as by ROCmFlashAttentionImpl (rocm_flash_attn.py:395-399)."""
        if attn_type is not None and getattr(attn_type, "name", str(attn_type)) != "DECODER":
            raise NotImplementedError("Encoder self-attention and encoder/decoder cross-attention are not implemented for "
                                      "MI355XAttentionImpl")
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
            # decided on the HOST (a `.any()` on the device tensor would sync the prefill stream once per layer and
            # cannot be captured): the builder's max_context_len when known, else the reference's rule
            # (rocm_flash_attn.py:455-459: cache present and block tables non-empty); the cached-context kernel
            # handles ctx_len == 0 per sequence.
            has_ctx = (key_cache is not None and prefill_meta.context_lens_tensor is not None
                       and prefill_meta.block_tables is not None and prefill_meta.block_tables.numel() > 0)
            if has_ctx and getattr(prefill_meta, "max_context_len", None) is not None:   # (ours; the reference's has no such field)
                has_ctx = prefill_meta.max_context_len > 0
            if has_ctx:
                # prefix-enabled attention (rocm_flash_attn.py:509-527)
                assert key_cache is not None
                output[:num_prefill_tokens] = PagedAttention.forward_prefix(
                    query, key, value, self.kv_cache_dtype, key_cache, value_cache,
                    prefill_meta.block_tables, prefill_meta.query_start_loc,
                    prefill_meta.seq_lens_tensor, prefill_meta.context_lens_tensor,
                    prefill_meta.max_query_len, self.alibi_slopes, self.sliding_window[0],
                    k_scale, v_scale,
                    # host-side maxima of the prefill sequences (seq_lens is a Python list): no .item() per layer
                    max_seq_len=max(prefill_meta.seq_lens), total_kv_tokens=sum(prefill_meta.seq_lens))
            else:
                out = ops.flash_attn_varlen(
                    query, key, value, prefill_meta.seq_start_loc,
                    prefill_meta.max_prefill_seq_len, self.scale, causal=True,
                    alibi_slopes=self.alibi_slopes, window_size=self.sliding_window)
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


# ---------------------------------------------------------------------------------------------------------------------
# AttentionState / AttentionMetadataBuilder (abstract.py:137-200): what the reference's model runner needs to SELECT this
# backend -- the persistent decode buffers of HIP-graph capture and the per-step metadata build.  Same behaviour as
# CommonAttentionState / CommonMetadataBuilder (attention/backends/utils.py:123-372), which ROCmFlashAttentionBackend
# returns from get_state_cls / get_builder_cls (rocm_flash_attn.py:41-47); restructured around numpy (one vectorised slot
# computation per sequence, one padded block-table array) instead of per-token Python lists.
# ---------------------------------------------------------------------------------------------------------------------
PAD_SLOT_ID = -1   # attention/backends/utils.py:18


class MI355XAttentionState:
    """Objects that live as long as the model runner: during HIP-graph capture the decode metadata points at
    persistent buffers (slot_mapping = -1, seq_lens = 1, the runner's graph_block_tables), which
    prepare_graph_input_buffers refreshes before every replay (utils.py:276-372)."""

    def __init__(self, runner):
        self.runner = runner
        self._is_graph_capturing = False

    @contextmanager
    def graph_capture(self, max_batch_size: int):
        self._is_graph_capturing = True
        dev = self.runner.device
        self._graph_slot_mapping = torch.full((max_batch_size, ), PAD_SLOT_ID, dtype=torch.long, device=dev)
        self._graph_seq_lens = torch.ones(max_batch_size, dtype=torch.int32, device=dev)
        self._graph_block_tables = torch.from_numpy(self.runner.graph_block_tables).to(device=dev)
        try:
            yield
        finally:
            self._is_graph_capturing = False
            del self._graph_slot_mapping
            del self._graph_seq_lens
            del self._graph_block_tables

    def graph_clone(self, batch_size: int) -> "MI355XAttentionState":
        assert self._is_graph_capturing
        return self.__class__(self.runner)

    def graph_capture_get_metadata_for_batch(self, batch_size: int, is_encoder_decoder_model: bool = False):
        assert self._is_graph_capturing
        if is_encoder_decoder_model:
            raise NotImplementedError("ROCm/HIP is not currently supported with encoder/decoder models.")
        return self.runner.attn_backend.make_metadata(
            num_prefills=0, num_prefill_tokens=0, num_decode_tokens=batch_size,
            slot_mapping=self._graph_slot_mapping[:batch_size], seq_lens=None,
            seq_lens_tensor=self._graph_seq_lens[:batch_size], max_query_len=None, max_prefill_seq_len=0,
            # the capture-time maximum: partition counts / grid.z are fixed at capture
            max_decode_seq_len=self.runner.max_seq_len_to_capture, query_start_loc=None, seq_start_loc=None,
            context_lens_tensor=None, block_tables=self._graph_block_tables[:batch_size], use_cuda_graph=True)

    def get_graph_input_buffers(self, attn_metadata, is_encoder_decoder_model: bool = False) -> Dict[str, Any]:
        return {"slot_mapping": attn_metadata.slot_mapping,
                "seq_lens_tensor": attn_metadata.decode_metadata.seq_lens_tensor,
                "block_tables": attn_metadata.decode_metadata.block_tables}

    def prepare_graph_input_buffers(self, input_buffers: Dict[str, Any], attn_metadata,
                                    is_encoder_decoder_model: bool = False) -> None:
        # (slot_mapping is refreshed by the runner itself, model_runner.py: CUDAGraphRunner.forward)
        input_buffers["seq_lens_tensor"].copy_(attn_metadata.decode_metadata.seq_lens_tensor, non_blocking=True)
        input_buffers["block_tables"].copy_(attn_metadata.decode_metadata.block_tables, non_blocking=True)

    def begin_forward(self, model_input) -> None:
        return


def _h2d(data, dtype: torch.dtype, device, pin_memory: bool) -> torch.Tensor:
    tcpu = torch.as_tensor(np.asarray(data), dtype=dtype)
    if pin_memory and torch.cuda.is_available():
        tcpu = tcpu.pin_memory()
    return tcpu.to(device=device, non_blocking=True)


class MI355XAttentionMetadataBuilder:
    """Per-step metadata from the runner's per-sequence-group data (CommonMetadataBuilder, utils.py:123-274):
    prefill sequences first, then decode tokens; slot = block_table[pos // block] * block + pos % block, -1 for
    profile runs, for tokens that fall out of a sliding window and for graph padding; under a captured graph the
    block tables are the runner's persistent [max_batch, max_blocks] array."""

    def __init__(self, input_builder) -> None:
        self.slot_chunks: List[np.ndarray] = []
        self.prefill_seq_lens: List[int] = []
        self.context_lens: List[int] = []
        self.block_tables: List[List[int]] = []
        self.curr_seq_lens: List[int] = []
        self.num_prefills = 0
        self.num_prefill_tokens = 0
        self.num_decode_tokens = 0
        self.input_builder = input_builder
        self.runner = input_builder.runner
        self.sliding_window = input_builder.sliding_window
        self.block_size = input_builder.block_size
        self.use_v2_block_manager = input_builder.scheduler_config.use_v2_block_manager

    def _slots(self, block_table, start: int, end: int) -> np.ndarray:
        pos = np.arange(start, end, dtype=np.int64)
        return np.asarray(block_table, dtype=np.int64)[pos // self.block_size] * self.block_size + pos % self.block_size

    def _add_seq_group(self, inter_data, chunked_prefill_enabled: bool) -> None:
        is_prompt = inter_data.is_prompt
        block_tables = inter_data.block_tables
        for (seq_id, token_len, seq_len, curr_seq_len, query_len, context_len, curr_sliding_window_block) in zip(
                inter_data.seq_ids, [len(t) for t in inter_data.input_tokens], inter_data.orig_seq_lens,
                inter_data.seq_lens, inter_data.query_lens, inter_data.context_lens,
                inter_data.curr_sliding_window_blocks):
            self.context_lens.append(context_len)
            if is_prompt:
                self.num_prefills += 1
                self.num_prefill_tokens += token_len
                self.prefill_seq_lens.append(seq_len)
            else:
                assert query_len == 1, f"seq_len: {seq_len}, context_len: {context_len}, query_len: {query_len}"
                self.num_decode_tokens += query_len
                self.curr_seq_lens.append(curr_seq_len)
            block_table: List[int] = []
            if inter_data.prefix_cache_hit:
                block_table = inter_data.computed_block_nums
            elif (chunked_prefill_enabled or not is_prompt) and block_tables is not None:
                block_table = block_tables[seq_id][-curr_sliding_window_block:]
            self.block_tables.append(block_table)
            # slot mapping (utils.py:40-121)
            profile_run = block_tables is None or (isinstance(block_tables, dict)
                                                   and all(v is None for v in block_tables.values()))
            if profile_run:
                self.slot_chunks.append(np.full(seq_len, PAD_SLOT_ID, dtype=np.int64))
                continue
            start_idx = 0
            if is_prompt and self.sliding_window is not None:
                assert self.use_v2_block_manager or context_len == 0, \
                    "Prefix caching is currently not supported with sliding window attention in V1 block manager"
                start_idx = max(0, query_len - self.sliding_window)
            pad = max(0, start_idx - context_len)
            if pad:
                self.slot_chunks.append(np.full(pad, PAD_SLOT_ID, dtype=np.int64))
            self.slot_chunks.append(self._slots(block_tables[seq_id], max(start_idx, context_len), seq_len))

    def build(self, seq_lens: List[int], query_lens: List[int], cuda_graph_pad_size: int, batch_size: int):
        for inter_data in self.input_builder.inter_data_list:
            self._add_seq_group(inter_data, self.input_builder.chunked_prefill_enabled)
        device = self.runner.device
        use_captured_graph = cuda_graph_pad_size != -1
        max_query_len = max(query_lens)
        assert max_query_len > 0, f"query_lens: {query_lens}"
        num_decode_tokens = self.num_decode_tokens
        slots = np.concatenate(self.slot_chunks) if self.slot_chunks else np.zeros(0, np.int64)
        if use_captured_graph:
            slots = np.concatenate([slots, np.full(cuda_graph_pad_size, PAD_SLOT_ID, dtype=np.int64)])
            num_decode_tokens = batch_size
            table = self.runner.graph_block_tables[:batch_size]      # persistent: padded rows keep their zeros
            for i, bt in enumerate(self.block_tables):
                if bt:
                    table[i, :len(bt)] = bt
            block_tables = torch.from_numpy(table).to(device, non_blocking=True)
        else:
            width = max((len(bt) for bt in self.block_tables), default=0)
            table = np.zeros((len(self.block_tables), width), dtype=np.int32)
            for i, bt in enumerate(self.block_tables):
                table[i, :len(bt)] = bt
            block_tables = torch.from_numpy(table).to(device)
        pin = getattr(self.runner, "pin_memory", False)
        seq_lens_tensor = _h2d(seq_lens, torch.int, device, pin)
        query_lens_tensor = _h2d(query_lens, torch.long, device, pin)
        query_start_loc = torch.zeros(len(query_lens) + 1, dtype=torch.int32, device=device)
        seq_start_loc = torch.zeros(len(seq_lens) + 1, dtype=torch.int32, device=device)
        torch.cumsum(seq_lens_tensor, dim=0, dtype=seq_start_loc.dtype, out=seq_start_loc[1:])
        torch.cumsum(query_lens_tensor, dim=0, dtype=query_start_loc.dtype, out=query_start_loc[1:])
        return MI355XAttentionMetadata(
            num_prefills=self.num_prefills, slot_mapping=_h2d(slots, torch.long, device, pin),
            num_prefill_tokens=self.num_prefill_tokens, num_decode_tokens=num_decode_tokens, seq_lens=seq_lens,
            seq_lens_tensor=seq_lens_tensor, max_query_len=max_query_len,
            max_prefill_seq_len=max(self.prefill_seq_lens, default=0),
            max_decode_seq_len=max(self.curr_seq_lens, default=0), query_start_loc=query_start_loc,
            seq_start_loc=seq_start_loc, context_lens_tensor=_h2d(self.context_lens, torch.int, device, pin),
            block_tables=block_tables, use_cuda_graph=use_captured_graph,
            max_context_len=max(self.context_lens[:self.num_prefills], default=0))
