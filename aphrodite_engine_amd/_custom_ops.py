"""Host-side mirror of the reference's ``aphrodite/_custom_ops.py`` for the hot
path: same function names, argument order, meaning and error behaviour, each
bound to the MI355X HIP kernels through the C ABI (no CPU fallback).

Reference: aphrodite/_custom_ops.py (paged_attention_v1 :85-115, v2 :118-151,
paged_attention_rocm :154-178, rms_norm/fused_add_rms_norm :192-199,
awq_dequantize/awq_gemm :223-240, gptq_gemm/gptq_shuffle :243-270,
cutlass_scaled_mm :497-517, scaled_fp8_quant :632-685, reshape_and_cache
:925-946, convert_fp8 :984-989).

Tensors are borrowed; ``Tensor!`` outputs are caller-allocated; returned
tensors are allocated here on the input's device.  Every launch goes to
``torch.cuda.current_stream()`` and nothing synchronises, so the ops can be
captured into a HIP graph exactly like the reference's.
"""
import contextlib
import os
from typing import List, Optional, Tuple

import torch

from .switches import switch
from . import _lib
from ._lib import check

FP8_DTYPE = torch.float8_e4m3fn  # gfx950 is OCP, not e4m3fnuz

_DT = {torch.float16: _lib.F16, torch.bfloat16: _lib.BF16, torch.float32: _lib.F32}
_KV = {"auto": _lib.KV_AUTO, "fp8": _lib.KV_FP8_E4M3, "fp8_e4m3": _lib.KV_FP8_E4M3,
       "fp8_e5m2": _lib.KV_FP8_E5M2}

_WS_BYTES = 128 << 20
_workspaces = {}


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def reload_env() -> None:
    """Have the library re-read its environment switches (csrc/common.h ``Knobs``): they are read once at load and never on
    a launch path, so a process that changes one afterwards says so."""
    _lib.lib().aphro_reload_env()


@contextlib.contextmanager
def knob(name: str, value):
    """``with ops.knob("APHRO_PA_SPLITS", 4): ...`` -- set (``None``: unset) one of the library's environment switches for the
    block: the variable is changed in this process, the library told to re-read, and both undone on exit."""
    old = os.environ.get(name)
    try:
        if value is None:
            os.environ.pop(name, None)
        else:
            os.environ[name] = str(value)
        reload_env()
        yield
    finally:
        if old is None:
            os.environ.pop(name, None)
        else:
            os.environ[name] = old
        reload_env()


def _dt(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise RuntimeError(f"Unsupported dtype {t.dtype}") from None


def _dt_of(dtype: torch.dtype) -> int:
    try:
        return _DT[dtype]
    except KeyError:
        raise RuntimeError(f"Unsupported dtype {dtype}") from None


def _kv(kv_cache_dtype: str) -> int:
    try:
        return _KV[kv_cache_dtype]
    except KeyError:
        raise RuntimeError(
            f"Unsupported data type of kv cache: {kv_cache_dtype}") from None


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _workspace(device: torch.device, nbytes: int) -> torch.Tensor:
    """Persistent split-K / stream-K scratch (stable address -> HIP-graph safe).  ONE buffer per device: ops that use it
    are ordered by the stream they are issued on -- the package issues all of them on the current stream (the side stream
    of distributed/overlap.py carries only the all-reduce).  A caller that runs these ops concurrently on several streams
    must give each stream its own scratch through the C ABI (every entry point takes workspace pointers)."""
    if nbytes > _WS_BYTES:
        return torch.empty(nbytes, dtype=torch.uint8, device=device)
    ws = _workspaces.get(device)
    if ws is None:
        ws = torch.empty(_WS_BYTES, dtype=torch.uint8, device=device)
        _workspaces[device] = ws
    return ws


_QUANT_SCRATCH = {}


def _quant_scratch(device: torch.device) -> torch.Tensor:
    """Per-device scratch of the atomic-free dynamic fp8 quantisation (2048 partial maxima)."""
    key = (device.type, device.index)
    t = _QUANT_SCRATCH.get(key)
    if t is None:
        t = torch.empty(2048, dtype=torch.float32, device=device)
        _QUANT_SCRATCH[key] = t
    return t


_FALLBACKS_SEEN = set()


def _library_fallback(op: str, why: str) -> None:
    """A quantised GEMM served by a LIBRARY GEMM (hipBLASLt through torch) instead of a hand-written kernel: legitimate
    (the reference does the same above 50 rows, q_gemm.cu:1529-1544; w8a8_utils.py:130-183 on ROCm) but never silent --
    one log line per (op, reason) per process."""
    key = (op, why)
    if key not in _FALLBACKS_SEEN:
        _FALLBACKS_SEEN.add(key)
        import logging
        logging.getLogger("aphrodite_engine_amd").warning("%s: library GEMM path (%s)", op, why)


def _require_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "aphrodite_engine_amd ops run on the MI355X only (got a CPU "
                "tensor); there is no CPU fallback")


# --------------------------------------------------------------------------
# paged attention
# --------------------------------------------------------------------------
def _paged_attention(out, exp_sums, max_logits, tmp_out, query, key_cache,
                     value_cache, num_kv_heads, scale, block_tables, seq_lens,
                     block_size, max_seq_len, alibi_slopes, kv_cache_dtype,
                     k_scale, v_scale, partition_size):
    _require_cuda(out, query, key_cache, value_cache, block_tables, seq_lens)
    num_seqs, num_heads, head_size = query.shape
    if query.stride(2) != 1 or query.stride(1) != head_size:
        raise RuntimeError("query heads must be contiguous")
    if not out.is_contiguous():
        raise RuntimeError("out must be contiguous")
    if block_tables.dtype != torch.int32 or seq_lens.dtype != torch.int32:
        raise RuntimeError("block_tables / seq_lens must be int32")
    if alibi_slopes is not None and alibi_slopes.dtype != torch.float32:
        alibi_slopes = alibi_slopes.float()
    check(_lib.lib().aphro_paged_attention(
        out.data_ptr(), _ptr(exp_sums), _ptr(max_logits), _ptr(tmp_out),
        query.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(),
        num_seqs, num_heads, num_kv_heads, head_size, float(scale),
        block_tables.data_ptr(), seq_lens.data_ptr(), block_tables.stride(0),
        block_size, int(max_seq_len), _ptr(alibi_slopes), query.stride(0),
        key_cache.stride(0), key_cache.stride(1), _dt(query), _kv(kv_cache_dtype),
        float(k_scale), float(v_scale), partition_size, _stream()),
        "paged_attention")


def paged_attention_v1(out, query, key_cache, value_cache, num_kv_heads, scale,
                       block_tables, seq_lens, block_size, max_seq_len,
                       alibi_slopes, kv_cache_dtype, k_scale, v_scale,
                       tp_rank: int = 0, blocksparse_local_blocks: int = 0,
                       blocksparse_vert_stride: int = 0,
                       blocksparse_block_size: int = 64,
                       blocksparse_head_sliding_step: int = 0) -> None:
    if blocksparse_vert_stride > 1:
        raise RuntimeError("blocksparse attention is out of scope (SURVEY 2.1)")
    _paged_attention(out, None, None, None, query, key_cache, value_cache,
                     num_kv_heads, scale, block_tables, seq_lens, block_size,
                     max_seq_len, alibi_slopes, kv_cache_dtype, k_scale, v_scale, 0)


def paged_attention_v2(out, exp_sum, max_logits, tmp_out, query, key_cache,
                       value_cache, num_kv_heads, scale, block_tables, seq_lens,
                       block_size, max_seq_len, alibi_slopes, kv_cache_dtype,
                       k_scale, v_scale, tp_rank: int = 0,
                       blocksparse_local_blocks: int = 0,
                       blocksparse_vert_stride: int = 0,
                       blocksparse_block_size: int = 64,
                       blocksparse_head_sliding_step: int = 0) -> None:
    if blocksparse_vert_stride > 1:
        raise RuntimeError("blocksparse attention is out of scope (SURVEY 2.1)")
    part = _partition_size(tmp_out, max_seq_len)
    _paged_attention(out, exp_sum, max_logits, tmp_out, query, key_cache,
                     value_cache, num_kv_heads, scale, block_tables, seq_lens,
                     block_size, max_seq_len, alibi_slopes, kv_cache_dtype,
                     k_scale, v_scale, part)


def paged_attention_rocm(out, exp_sum, max_logits, tmp_out, query, key_cache,
                         value_cache, num_kv_heads, scale, block_tables,
                         seq_lens, block_size, max_seq_len, alibi_slopes,
                         kv_cache_dtype, k_scale, v_scale) -> None:
    """`_rocm_C::paged_attention` (kernels/rocm/torch_bindings.cpp): the reference's ROCm backend calls it for every decode
    batch on gfx9 (rocm_flash_attn.py:539-560, 633-641) with scratch for 512-token partitions.  Only ``out`` is read back by
    the caller; where one launch over whole sequences is the faster form -- the reference's own v1 / v2 rule,
    paged_attn.py:112-121: max_seq_len <= 8192 and (one partition or num_seqs * num_heads > 512) -- the op runs that form and
    leaves the scratch untouched (25.4 against 26.4 + 5.0 us for the partition kernel + its reduce launch at configs[1]).
    APHRO_PA_ROCM_PARTITIONED=1 keeps the partitioned form (and the exp_sums / max_logits / tmp_out contract)."""
    part = _partition_size(tmp_out, max_seq_len)
    num_seqs, num_heads = query.shape[0], query.shape[1]
    if (max_seq_len <= 8192 and ((max_seq_len + part - 1) // part == 1 or num_seqs * num_heads > 512)
            and not switch("APHRO_PA_ROCM_PARTITIONED")):
        part, exp_sum, max_logits, tmp_out = 0, None, None, None
    _paged_attention(out, exp_sum, max_logits, tmp_out, query, key_cache,
                     value_cache, num_kv_heads, scale, block_tables, seq_lens,
                     block_size, max_seq_len, alibi_slopes, kv_cache_dtype,
                     k_scale, v_scale, part)


def _partition_size(tmp_out: torch.Tensor, max_seq_len: int) -> int:
    """Callers size the scratch for 512-token partitions
    (paged_attn.py:13,118-119; rocm_flash_attn.py:23,544-546)."""
    part = 512
    need = (max_seq_len + part - 1) // part
    if tmp_out.shape[2] < need:
        raise RuntimeError(
            f"tmp_out has {tmp_out.shape[2]} partitions, need {need} for "
            f"max_seq_len={max_seq_len}")
    if not (tmp_out.is_contiguous()):
        raise RuntimeError("tmp_out must be contiguous")
    return part


def flash_attn_varlen(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor,
                      cu_seqlens: torch.Tensor, max_seqlen: int, softmax_scale: float,
                      causal: bool = True,
                      alibi_slopes: Optional[torch.Tensor] = None,
                      window_size: Optional[tuple] = None) -> torch.Tensor:
    """Prefill self-attention over packed sequences: the role of
    triton_attention / flash_attn_varlen_func in ROCmFlashAttentionImpl
    (rocm_flash_attn.py:455-508).  q [T,Hq,hd], k/v [T,Hkv,hd] (token-strided
    views allowed), cu_seqlens int32 [B+1]; returns [T,Hq,hd].  ``window_size`` = flash_attn_varlen_func's (left, right)
    (rocm_flash_attn.py:506; (-1, -1) / None = off): under causal attention query i sees keys i - left .. i."""
    _require_cuda(q, k, v, cu_seqlens)
    t, hq, hd = q.shape
    hkv = k.shape[1]
    for x in (q, k, v):
        if x.stride(2) != 1 or x.stride(1) != hd:
            raise RuntimeError("flash_attn_varlen: heads must be contiguous")
    if cu_seqlens.dtype != torch.int32:
        cu_seqlens = cu_seqlens.to(torch.int32)
    out = torch.empty((t, hq, hd), dtype=q.dtype, device=q.device)
    if alibi_slopes is not None and alibi_slopes.dtype != torch.float32:
        alibi_slopes = alibi_slopes.float()
    left = int(window_size[0]) if window_size is not None else -1
    if left >= 0:
        if not causal:
            raise RuntimeError("flash_attn_varlen: a sliding window needs causal attention")
        check(_lib.lib().aphro_flash_attn_varlen_window(
            out.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr(),
            cu_seqlens.data_ptr(), cu_seqlens.numel() - 1, int(max_seqlen), hq, hkv, hd,
            q.stride(0), k.stride(0), v.stride(0), float(softmax_scale),
            1, _ptr(alibi_slopes), left + 1, _dt(q), _stream()),
            "flash_attn_varlen")
        return out
    check(_lib.lib().aphro_flash_attn_varlen(
        out.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr(),
        cu_seqlens.data_ptr(), cu_seqlens.numel() - 1, int(max_seqlen), hq, hkv, hd,
        q.stride(0), k.stride(0), v.stride(0), float(softmax_scale),
        1 if causal else 0, _ptr(alibi_slopes), _dt(q), _stream()),
        "flash_attn_varlen")
    return out


# context_attention_fwd without host-side length hints: a workspace bound of new tokens + block-table capacity per sequence
# is used without a device sync while it stays below this many bytes
CONTEXT_ATTN_NOSYNC_WS_BYTES = 256 << 20
CONTEXT_ATTN_GATHER_MIN_LEN = 1024     # (kept for callers that imported it: the third-generation kernel's threshold)


def context_attention_fwd(q, k, v, o, kv_cache_dtype: str, k_cache, v_cache, b_loc, b_start_loc,
                          b_seq_len, b_ctx_len, max_input_len: int, k_scale: float = 1.0,
                          v_scale: float = 1.0, alibi_slopes: Optional[torch.Tensor] = None,
                          sliding_window: Optional[int] = None, max_seq_len: Optional[int] = None,
                          total_kv_tokens: Optional[int] = None) -> None:
    """attention/ops/prefix_prefill.py:696-711 (same argument order; ``max_seq_len`` / ``total_kv_tokens`` are optional
    host-side hints -- max and sum of ``b_seq_len`` -- that save a device sync on the long-prompt path).  q/k/v/o [T,H,hd]
    views of the new tokens; k_cache [NB,Hkv,hd/x,block,x], v_cache [NB,Hkv,hd,block];
    b_loc = block tables, b_start_loc = query_start_loc [B+1], b_seq_len = context + new,
    b_ctx_len = cached context.  Like the reference the softmax scale is 1/sqrt(hd)
    (prefix_prefill.py:745)."""
    _require_cuda(q, k, v, o, k_cache, v_cache, b_loc, b_start_loc, b_seq_len, b_ctx_len)
    if q.dtype not in (torch.float16, torch.bfloat16):
        raise RuntimeError("context_attention_fwd: query must be fp16 or bf16")
    head_size = q.shape[-1]
    if q.stride(1) != head_size or k.stride(1) != head_size or v.stride(1) != head_size \
            or o.stride(1) != head_size:
        raise RuntimeError("context_attention_fwd: heads must be contiguous")
    batch = b_seq_len.shape[0]
    i32 = lambda t_: t_ if t_.dtype == torch.int32 else t_.to(torch.int32)
    b_loc, b_start_loc, b_seq_len, b_ctx_len = map(i32, (b_loc, b_start_loc, b_seq_len, b_ctx_len))
    b_loc = b_loc.contiguous()
    slopes = None
    if alibi_slopes is not None:
        slopes = alibi_slopes.to(device=q.device, dtype=torch.float32).contiguous()
    win = int(sliding_window) if sliding_window is not None and sliding_window > 0 else 0
    lib = _lib.lib()
    if head_size in (64, 96, 128, 256) and not switch("APHRO_CA_NO_GATHER"):
        # gather the cached context once, then the prefill tile machines over context + new tokens (every head size, ALiBi,
        # sliding window: round 3).  The caller passes the host-side maxima when it has them (the attention metadata does);
        # without them a host-side bound -- new tokens + block-table capacity per sequence -- sizes the workspace when that
        # stays small, else one sync.
        if max_seq_len is not None and total_kv_tokens is not None:
            msl, tot = int(max_seq_len), int(total_kv_tokens)
        else:
            cap = int(max_input_len) + b_loc.shape[1] * v_cache.shape[3]
            if batch * cap * k.shape[1] * head_size * 4 <= CONTEXT_ATTN_NOSYNC_WS_BYTES:
                msl, tot = cap, batch * cap
            else:
                msl, tot = int(b_seq_len.max().item()), int(b_seq_len.sum().item())
        ws = _workspace(q.device, lib.aphro_context_attention_workspace_bytes(tot, batch, k.shape[1], head_size))
        check(lib.aphro_context_attention_gathered(
            o.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(),
            b_loc.data_ptr(), b_start_loc.data_ptr(), b_seq_len.data_ptr(), b_ctx_len.data_ptr(), batch,
            int(max_input_len), msl, tot, b_loc.shape[1], q.shape[1], k.shape[1], head_size, v_cache.shape[3],
            k_cache.shape[4], q.stride(0), k.stride(0), v.stride(0), o.stride(0), head_size ** -0.5,
            float(k_scale), float(v_scale), _ptr(slopes), win, _dt(q), _kv(kv_cache_dtype), ws.data_ptr(), ws.numel(),
            _stream()), "context_attention_fwd")
        return
    # first-round kernel (scalar cache gathers): kept for head sizes the tile machines do not serve and for A/B runs
    check(lib.aphro_context_attention(
        o.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(),
        b_loc.data_ptr(), b_start_loc.data_ptr(), b_seq_len.data_ptr(), b_ctx_len.data_ptr(), batch,
        int(max_input_len), b_loc.shape[1], q.shape[1], k.shape[1], head_size, v_cache.shape[3],
        k_cache.shape[4], q.stride(0), k.stride(0), v.stride(0), o.stride(0), head_size ** -0.5,
        float(k_scale), float(v_scale), _ptr(slopes), win, _dt(q), _kv(kv_cache_dtype), _stream()),
        "context_attention_fwd")


# --------------------------------------------------------------------------
# cache ops
# --------------------------------------------------------------------------
def reshape_and_cache(key, value, key_cache, value_cache, slot_mapping,
                      kv_cache_dtype: str, k_scale: float, v_scale: float) -> None:
    _require_cuda(key, value, key_cache, value_cache, slot_mapping)
    num_tokens, num_kv_heads, head_size = key.shape
    block_size, x = key_cache.shape[3], key_cache.shape[4]
    if slot_mapping.dtype != torch.int64:
        raise RuntimeError("slot_mapping must be int64")
    if key.stride(1) != head_size or value.stride(1) != head_size:
        raise RuntimeError("key/value heads must be contiguous")
    check(_lib.lib().aphro_reshape_and_cache(
        key.data_ptr(), value.data_ptr(), key_cache.data_ptr(),
        value_cache.data_ptr(), slot_mapping.data_ptr(), num_tokens,
        num_kv_heads, head_size, block_size, x, key.stride(0), value.stride(0),
        _dt(key), _kv(kv_cache_dtype), float(k_scale), float(v_scale),
        _stream()), "reshape_and_cache")


def reshape_and_cache_flash(key, value, key_cache, value_cache, slot_mapping,
                            kv_cache_dtype: str, k_scale: float, v_scale: float) -> None:
    """_custom_ops.py reshape_and_cache_flash: caches in the [NB, block, H, hd] layout."""
    _require_cuda(key, value, key_cache, value_cache, slot_mapping)
    num_tokens, num_heads, head_size = key.shape
    block_size = key_cache.shape[1]
    if slot_mapping.dtype != torch.int64:
        raise RuntimeError("slot_mapping must be int64")
    if key_cache.stride(0) != value_cache.stride(0):
        raise RuntimeError("key_cache and value_cache must have the same block stride")
    if key.stride(1) != head_size or value.stride(1) != head_size:
        raise RuntimeError("key/value heads must be contiguous")
    check(_lib.lib().aphro_reshape_and_cache_flash(
        key.data_ptr(), value.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(),
        slot_mapping.data_ptr(), num_tokens, num_heads, head_size, block_size,
        key_cache.stride(0), key.stride(0), value.stride(0), _dt(key), _kv(kv_cache_dtype),
        float(k_scale), float(v_scale), _stream()), "reshape_and_cache_flash")


def copy_blocks(key_caches: List[torch.Tensor], value_caches: List[torch.Tensor],
                block_mapping: torch.Tensor) -> None:
    """_C_cache_ops::copy_blocks (cache_kernels.cu:103-148): for every layer copy block
    src -> dst for each (src, dst) row of block_mapping (int64 [num_pairs, 2], device)."""
    num_layers = len(key_caches)
    if num_layers != len(value_caches):
        raise RuntimeError("copy_blocks: key_caches and value_caches differ in length")
    if num_layers == 0 or block_mapping.numel() == 0:
        return
    _require_cuda(key_caches[0], value_caches[0], block_mapping)
    dev = key_caches[0].device
    kp = torch.tensor([t.data_ptr() for t in key_caches], dtype=torch.int64, device=dev)
    vp = torch.tensor([t.data_ptr() for t in value_caches], dtype=torch.int64, device=dev)
    bm = block_mapping.to(torch.int64).contiguous()
    block_bytes = key_caches[0][0].numel() * key_caches[0].element_size()
    check(_lib.lib().aphro_copy_blocks(kp.data_ptr(), vp.data_ptr(), num_layers, bm.data_ptr(),
                                       bm.shape[0], block_bytes, _stream()), "copy_blocks")


def swap_blocks(src: torch.Tensor, dst: torch.Tensor, block_mapping: torch.Tensor) -> None:
    """_C_cache_ops::swap_blocks (cache_kernels.cu:24-63); block_mapping is a CPU int64
    [num_pairs, 2] tensor, src/dst are whole caches indexed by block along dim 0."""
    if block_mapping.device.type != "cpu":
        raise RuntimeError("block_mapping must be on CPU")
    sd, dd = src.device.type, dst.device.type
    if sd == "cuda" and dd == "cuda":
        if src.device.index != dst.device.index:
            raise RuntimeError("src and dst must be on the same GPU")
        kind = 0
    elif sd == "cuda" and dd == "cpu":
        kind = 1
    elif sd == "cpu" and dd == "cuda":
        kind = 2
    else:
        raise RuntimeError("Invalid device combination")
    bm = block_mapping.to(torch.int64).contiguous()
    block_bytes = src[0].numel() * src.element_size()
    if bm.numel() == 0:
        return
    with torch.cuda.device(src.device if sd == "cuda" else dst.device):
        check(_lib.lib().aphro_swap_blocks(src.data_ptr(), dst.data_ptr(), bm.data_ptr(),
                                           bm.shape[0], block_bytes, kind, _stream()), "swap_blocks")


def convert_fp8(output: torch.Tensor, input: torch.Tensor, scale: float = 1.0,
                kv_dtype: str = "fp8") -> None:
    _require_cuda(output, input)
    # cache_kernels.cu:373-409 accepts "auto", "fp8" and "fp8_e4m3"; "auto" between a 16 / 32-bit tensor and a uint8 cache
    # can only mean the platform's fp8 format (the reference instantiates kAuto there, whose scaled_convert is an assert):
    # e4m3.  fp8_e5m2 is accepted as an extension.
    kvd = _kv("fp8_e4m3" if kv_dtype == "auto" else kv_dtype)
    if not (output.is_contiguous() and input.is_contiguous()):
        raise RuntimeError("convert_fp8 needs contiguous tensors")
    to_fp8 = output.dtype == torch.uint8
    hp = input if to_fp8 else output
    check(_lib.lib().aphro_convert_fp8(
        output.data_ptr(), input.data_ptr(), input.numel(), float(scale),
        _dt(hp), kvd, 1 if to_fp8 else 0, _stream()), "convert_fp8")


# --------------------------------------------------------------------------
# GPTQ
# --------------------------------------------------------------------------
def gptq_shuffle(q_weight: torch.Tensor, q_perm: torch.Tensor, bit: int) -> None:
    _require_cuda(q_weight)
    rows, n = q_weight.shape
    perm = q_perm if (q_perm is not None and q_perm.numel() > 0) else None
    if bit in (2, 3, 8):
        # the layout after the shuffle is private to the kernels that read it: these widths keep the checkpoint's
        # sequential bitstring, act-order rows made sequential (csrc/wnx_gemm.hip)
        if perm is not None:
            tmp = torch.empty_like(q_weight)
            check(_lib.lib().aphro_gptq_make_sequential_bits(q_weight.data_ptr(), tmp.data_ptr(),
                                                             perm.to(torch.int32).data_ptr(), rows * 32 // bit, n, bit,
                                                             _stream()), "gptq_shuffle")
            q_weight.copy_(tmp)
        return
    tmp = torch.empty_like(q_weight) if perm is not None else None
    if perm is not None and perm.dtype != torch.int32:
        perm = perm.to(torch.int32)
    check(_lib.lib().aphro_gptq_shuffle(
        q_weight.data_ptr(), _ptr(perm), rows * (32 // bit), n, bit, _ptr(tmp),
        _stream()), "gptq_shuffle")


def gptq_dequant(b_q_weight, b_gptq_qzeros, b_gptq_scales, b_g_idx,
                 use_exllama: bool, bit: int = 4, zero_offset: int = 1):
    """temp_dq of q_gemm.cu:1520-1535: the fp16 weight the reference hands to
    hipBLAS for M > 50."""
    if bit in (2, 3, 8):
        k, n = b_q_weight.shape[0] * 32 // bit, b_q_weight.shape[1]
        out = torch.empty((k, n), dtype=b_gptq_scales.dtype, device=b_q_weight.device)
        g_idx = b_g_idx.to(torch.int32) if (not use_exllama and b_g_idx is not None and b_g_idx.numel() > 0) else None
        check(_lib.lib().aphro_gptq_dequant_bits(b_q_weight.data_ptr(), b_gptq_qzeros.data_ptr(), b_gptq_scales.data_ptr(),
                                                 _ptr(g_idx), out.data_ptr(), k, n, b_gptq_scales.shape[0], bit,
                                                 _dt(b_gptq_scales), _stream()), "gptq_dequant")
        return out
    if bit != 4:
        raise RuntimeError(f"GPTQ weight width {bit} is not one of 2, 3, 4, 8")
    k, n = b_q_weight.shape[0] * 8, b_q_weight.shape[1]
    out = torch.empty((k, n), dtype=b_gptq_scales.dtype, device=b_q_weight.device)
    g_idx = None
    if not use_exllama and b_g_idx is not None and b_g_idx.numel() > 0:
        g_idx = b_g_idx.to(torch.int32)
    check(_lib.lib().aphro_gptq_dequant(
        b_q_weight.data_ptr(), b_gptq_qzeros.data_ptr(), b_gptq_scales.data_ptr(),
        _ptr(g_idx), out.data_ptr(), k, n, b_gptq_scales.shape[0],
        1 if use_exllama else 0, zero_offset, _dt(b_gptq_scales), _stream()),
        "gptq_dequant")
    return out


# M above which the MFMA-bound prefill kernel (csrc/wna16_gemm_large.hip: dequant in registers, 32x32x16 MFMA,
# activations through direct-to-LDS loads) takes over from the HBM-bound decode kernel
WNA16_LARGE_MIN_M = 64


def wna16_large_ok(m: int, n: int, k: int, groups: int) -> bool:
    gs = k // max(groups, 1)
    return m > WNA16_LARGE_MIN_M and n % 128 == 0 and k % 64 == 0 and groups > 0 and k % groups == 0 and gs % 64 == 0 \
        and (k // 8) * n * 4 < 2 ** 32 and m * k * 2 < 2 ** 32


def wna16_prefers_large(m: int, n: int, k: int) -> bool:
    """Between 65 and 128 rows the decode kernel (two 64-row passes over the weights) still beats the tile machine on
    the small projections (weights < 16 MiB: qkv 21.6 vs 27.4 us, o 20.0 vs 23.8 us at M = 96), not on the large ones
    (gate_up 59 vs 52 us, down 43 vs 35 us) -- tools/mid_gemm_bench.py, profiles/r2_mid_gemm.txt."""
    return m > 128 or n * k >= 2 ** 25


def _wna16_large(a, qweight, qzeros, scales, perm, zero_offset):
    m, k = a.shape
    n = qweight.shape[1]
    lib = _lib.lib()
    if perm is not None:
        a = a[:, perm.long()]                       # act-order: gather once (q_gemm.cu:219-226)
    if a.stride(1) != 1 or a.stride(0) % 8 != 0 or a.data_ptr() % 16 != 0:
        a = a.contiguous()
    out = torch.empty((m, n), dtype=a.dtype, device=a.device)
    nbytes = lib.aphro_wna16_gemm_large_workspace_bytes(m, n, k, scales.shape[0], _dt(a))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=a.device) if nbytes else None
    check(lib.aphro_wna16_gemm_large(a.data_ptr(), qweight.data_ptr(), qzeros.data_ptr(), scales.data_ptr(),
                                     out.data_ptr(), _ptr(ws), nbytes, m, n, k, scales.shape[0], a.stride(0),
                                     zero_offset, _dt(a), _stream()), "wna16_gemm_large")
    return out


def wna16_gemm_large_silu_supported(m: int, n: int, k: int, groups: int) -> bool:
    return bool(_lib.lib().aphro_wna16_gemm_large_silu_supported(m, n, k, groups)) and m * k * 2 < 2 ** 32


def wna16_gemm_large_silu(a: torch.Tensor, qweight: torch.Tensor, qzeros: torch.Tensor, scales: torch.Tensor,
                          zero_offset: int) -> torch.Tensor:
    """Prompt-sized gate_up GEMM on INTERLEAVED (gate_j, up_j) columns (ops.interleave_gate_up) with SiluAndMul in the
    epilogue: act [M, N / 2] -- the bits of the GEMM followed by silu_and_mul(..., interleaved=True), without the [M, N]
    round trip through HBM.  K-packed exllama weights, no act-order (csrc/wna16_gemm_large.hip)."""
    _require_cuda(a, qweight, qzeros, scales)
    m, k = a.shape
    n = qweight.shape[1]
    lib = _lib.lib()
    if a.stride(1) != 1 or a.stride(0) % 8 != 0 or a.data_ptr() % 16 != 0:
        a = a.contiguous()
    out = torch.empty((m, n // 2), dtype=a.dtype, device=a.device)
    nbytes = lib.aphro_wna16_gemm_large_workspace_bytes(m, n, k, scales.shape[0], _dt(a))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=a.device) if nbytes else None
    check(lib.aphro_wna16_gemm_large_silu(a.data_ptr(), qweight.data_ptr(), qzeros.data_ptr(), scales.data_ptr(),
                                          out.data_ptr(), _ptr(ws), nbytes, m, n, k, scales.shape[0], a.stride(0),
                                          zero_offset, _dt(a), _stream()), "wna16_gemm_large_silu")
    return out


def wna16_strip_unrelayout(strip: torch.Tensor, m: int, groups: int) -> torch.Tensor:
    """wna16_strip_relayout backwards: the strip-major copy (laid out for the M class ``m``) -> [K/8, N] exllama-ordered words.
    For the kernels that do not read the strip-major order when it is the only resident copy of the matrix."""
    _require_cuda(strip)
    out = torch.empty_like(strip)
    check(_lib.lib().aphro_wna16_strip_unrelayout(strip.data_ptr(), out.data_ptr(), m, strip.shape[1], strip.shape[0] * 8,
                                                  groups, _stream()), "wna16_strip_unrelayout")
    return out


def wna16_gemm_large_strip(a: torch.Tensor, strip: torch.Tensor, qzeros: torch.Tensor, scales: torch.Tensor,
                           zero_offset: int, silu: bool = False, strip_m: int = 32) -> torch.Tensor:
    """_wna16_large / wna16_gemm_large_silu on the STRIP-MAJOR copy of the weights (wna16_strip_relayout(qweight, strip_m,
    groups)): every plan of the tile machine addresses its 16-byte pieces in place.
    Same bits as the [K/8, N] entries (csrc/wna16_gemm_large.hip, Wna16LargeParams::strip)."""
    _require_cuda(a, strip, qzeros, scales)
    m, k = a.shape
    n = strip.shape[1]
    lib = _lib.lib()
    if a.stride(1) != 1 or a.stride(0) % 8 != 0 or a.data_ptr() % 16 != 0:
        a = a.contiguous()
    out = torch.empty((m, n // 2 if silu else n), dtype=a.dtype, device=a.device)
    nbytes = lib.aphro_wna16_gemm_large_strip_workspace_bytes(m, n, k, scales.shape[0], _dt(a), strip_m)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=a.device) if nbytes else None
    check(lib.aphro_wna16_gemm_large_strip(a.data_ptr(), strip.data_ptr(), qzeros.data_ptr(), scales.data_ptr(),
                                           out.data_ptr(), _ptr(ws), nbytes, m, n, k, scales.shape[0], a.stride(0),
                                           zero_offset, _dt(a), 1 if silu else 0, strip_m, _stream()), "wna16_gemm_large_strip")
    return out


def wna16_linear_strip(a: torch.Tensor, strip: torch.Tensor, qzeros: torch.Tensor, scales: torch.Tensor,
                       zero_offset: int, strip_m: int = 32) -> torch.Tensor:
    """[M, N] = a . dequant(W) for ANY M when the strip-major copy is the only resident copy of W (model.enable_one_copy):
    <= 32 rows the one-launch decode GEMM; prompt-sized M the tile machine reading the strip-major order in place; 33..64 rows
    on the big matrices the one-pass kernel at strip-major addresses; else 32-row passes of the decode GEMM (f16) -- and where
    none of these serves the call, the [K/8, N] order rebuilt into a transient (one more pass over the packed weights) for
    the generic kernels."""
    _require_cuda(a, strip, qzeros, scales)
    m, k = a.shape
    n, groups = scales.shape[1], scales.shape[0]
    if m == 0:
        return torch.empty((0, n), dtype=a.dtype, device=a.device)
    large = wna16_large_ok(m, n, k, groups) and not switch("APHRO_WNA16_NO_LARGE")
    if large and wna16_prefers_large(m, n, k):
        return wna16_gemm_large_strip(a, strip, qzeros, scales, zero_offset, strip_m=strip_m)
    if wna16_prefers_mid(m, n, k) and wna16_mid_ok(m, n, k, groups) and a.dtype == scales.dtype and not switch("APHRO_WNA16_NO_MID"):
        # 33..64 rows on the big matrices: ONE pass of the 32x32x16 kernel over the strip-major words (the generic op's own rule)
        return wna16_gemm_mid_packed(wna16_pack_a(a), m, k, strip, qzeros, scales, zero_offset, strip_m=strip_m)
    if m <= 128:
        def rows32(x):          # <= 32 rows: the one-launch row-major GEMM (f16), else pack + the resident kernel (bf16 too)
            if x.dtype == torch.float16 and wna16_gemm_rowmajor_supported(x.shape[0], n, k, groups, x.dtype):
                return wna16_gemm_rowmajor(x, strip, qzeros, scales, zero_offset, strip_layout=True)
            ks = wna16_resident_ksplit(x.shape[0], n, k, groups) if x.dtype == scales.dtype else 0
            if ks <= 0:
                return None
            packed = wna16_pack_a(x)
            if ks == 1:
                return wna16_gemm_resident(packed, x.shape[0], k, strip, qzeros, scales, zero_offset, mode="out", strip_layout=True)
            slabs, _ = wna16_gemm_resident(packed, x.shape[0], k, strip, qzeros, scales, zero_offset, mode="slabs", strip_layout=True)
            return slabs.sum(0).to(x.dtype)
        parts = []
        for m0 in range(0, m, 32):
            part = rows32(a[m0:m0 + 32])
            if part is None:
                break
            parts.append(part)
        else:
            return parts[0] if len(parts) == 1 else torch.cat(parts, 0)
    if large:
        return wna16_gemm_large_strip(a, strip, qzeros, scales, zero_offset, strip_m=strip_m)
    qweight = wna16_strip_unrelayout(strip, strip_m, groups)
    if m >= GPTQ_DEQUANT_MIN_M:         # gptq_gemm's own rule where the tile machine does not take the call
        _library_fallback("wna16_linear_strip", f"M={m} N={n} K={k}: " + ("APHRO_WNA16_NO_LARGE is set" if switch("APHRO_WNA16_NO_LARGE")
                          else "shape not tiled by wna16_gemm_large") + " (strip_unrelayout + gptq_dequant + matmul)")
        return torch.matmul(a, gptq_dequant(qweight, qzeros, scales, None, True, 4, zero_offset))
    return _wna16(a, qweight, qzeros, scales, None, zero_offset)


def wna16_mid_ok(m: int, n: int, k: int, groups: int) -> bool:
    return bool(_lib.lib().aphro_wna16_gemm_mid_supported(m, n, k, groups)) and m * k * 2 < 2 ** 32


def wna16_prefers_mid(m: int, n: int, k: int) -> bool:
    """33..64 rows: one pass of the 32x32x16 MFMA kernel over the weights against two passes of the decode kernel.
    Measured (tools/mid_gemm_bench.py, profiles/r2_mid_gemm.txt, whole op incl. activation pack / slab reduce): a win
    where the weight stream dominates the fixed costs -- gate_up 27.0 vs 37.7 us at M = 64 (25.1 vs 32.8 at 33), down
    23.7 vs 25.3; the small projections go the other way (qkv 21.5 vs 11.4, o 16.1 vs 11.7)."""
    return 32 < m <= 64 and n * k >= 2 ** 25


def _wna16_mid(a, qweight, qzeros, scales, perm, zero_offset):
    """Decode batches of 33..64 rows in ONE pass over the weights (csrc/wna16_gemm_mid.hip)."""
    m, k = a.shape
    n = qweight.shape[1]
    lib = _lib.lib()
    if perm is not None:
        a = a[:, perm.long()]                       # act-order: gather once (q_gemm.cu:219-226)
    if a.stride(1) != 1 or a.stride(0) % 8 != 0 or a.data_ptr() % 16 != 0:
        a = a.contiguous()
    out = torch.empty((m, n), dtype=a.dtype, device=a.device)
    nbytes = lib.aphro_wna16_gemm_mid_workspace_bytes(m, n, k, scales.shape[0])
    ws = torch.empty(nbytes, dtype=torch.uint8, device=a.device)
    check(lib.aphro_wna16_gemm_mid(a.data_ptr(), qweight.data_ptr(), qzeros.data_ptr(), scales.data_ptr(),
                                   out.data_ptr(), ws.data_ptr(), nbytes, m, n, k, scales.shape[0], a.stride(0),
                                   zero_offset, _dt(a), _stream()), "wna16_gemm_mid")
    return out


def _wna16(a, qweight, qzeros, scales, perm, zero_offset):
    m, k = a.shape
    n = qweight.shape[1]
    if wna16_large_ok(m, n, k, scales.shape[0]) and wna16_prefers_large(m, n, k) \
            and not switch("APHRO_WNA16_NO_LARGE"):
        return _wna16_large(a, qweight, qzeros, scales, perm, zero_offset)
    if wna16_prefers_mid(m, n, k) and wna16_mid_ok(m, n, k, scales.shape[0]) and not switch("APHRO_WNA16_NO_MID"):
        return _wna16_mid(a, qweight, qzeros, scales, perm, zero_offset)
    lib = _lib.lib()
    out = torch.empty((m, n), dtype=a.dtype, device=a.device)
    if a.stride(1) != 1:
        a = a.contiguous()
    ws = _workspace(a.device, lib.aphro_wna16_workspace_bytes(min(m, 64), n, k))
    a_tmp = torch.empty((min(m, 64), k), dtype=a.dtype, device=a.device) \
        if perm is not None else None
    for m0 in range(0, m, 64):
        rows = min(64, m - m0)
        a_blk = a[m0:m0 + rows]
        check(lib.aphro_gptq_gemm(
            a_blk.data_ptr(), qweight.data_ptr(), qzeros.data_ptr(),
            scales.data_ptr(), _ptr(perm), _ptr(a_tmp), out[m0:].data_ptr(),
            ws.data_ptr(), ws.numel(), rows, n, k, scales.shape[0], a.stride(0),
            zero_offset, _dt(a), _stream()), "gptq_gemm")
    return out


# M above which the weight is reconstructed once and a library GEMM is used;
# the reference switches at 50 rows (q_gemm.cu:28,1529-1544).
GPTQ_DEQUANT_MIN_M = 256
SCALED_MM_LIBRARY_MIN_M = 64   # above this the fp8 GEMM goes to hipBLASLt (torch._scaled_mm)


def gptq_gemm(a: torch.Tensor, b_q_weight: torch.Tensor,
              b_gptq_qzeros: torch.Tensor, b_gptq_scales: torch.Tensor,
              b_g_idx: torch.Tensor, use_exllama: bool, bit: int) -> torch.Tensor:
    _require_cuda(a, b_q_weight, b_gptq_qzeros, b_gptq_scales)
    if a.dtype != b_gptq_scales.dtype:
        raise RuntimeError("gptq_gemm: activations and scales must share a dtype")
    if bit in (2, 3, 8):
        return _gptq_gemm_bits(a, b_q_weight, b_gptq_qzeros, b_gptq_scales, b_g_idx, use_exllama, bit)
    if bit != 4:
        raise RuntimeError(f"gptq_gemm: weight width {bit} is not one of 2, 3, 4, 8")
    m = a.shape[0]
    large = use_exllama and wna16_large_ok(m, b_q_weight.shape[1], a.shape[1], b_gptq_scales.shape[0]) \
        and wna16_prefers_large(m, b_q_weight.shape[1], a.shape[1]) and not switch("APHRO_WNA16_NO_LARGE")
    if not use_exllama or (m >= GPTQ_DEQUANT_MIN_M and not large):
        w = gptq_dequant(b_q_weight, b_gptq_qzeros, b_gptq_scales, b_g_idx,
                         use_exllama, bit)
        if use_exllama and b_g_idx is not None and b_g_idx.numel() > 0:
            a = a[:, b_g_idx.long()]
        _library_fallback("gptq_gemm", "not exllama-shuffled" if not use_exllama else
                          f"M={m} N={b_q_weight.shape[1]} K={a.shape[1]}: " +
                          ("APHRO_WNA16_NO_LARGE is set" if switch("APHRO_WNA16_NO_LARGE") else
                           "shape not tiled by wna16_gemm_large") + " (gptq_dequant + matmul)")
        return torch.matmul(a, w)
    perm = None
    if b_g_idx is not None and b_g_idx.numel() > 0:
        perm = b_g_idx.to(torch.int32) if b_g_idx.dtype != torch.int32 else b_g_idx
    return _wna16(a, b_q_weight, b_gptq_qzeros, b_gptq_scales, perm, 1)


def _gptq_gemm_bits(a, qweight, qzeros, scales, g_idx, use_exllama: bool, bit: int) -> torch.Tensor:
    """2 / 3 / 8-bit GPTQ (q_gemm.cu:329-700, 1526-1560): the MFMA small-M kernel on the sequential layout at <= 32 rows
    (64 in two passes), else the weight reconstructed once (bit for bit the reference's reconstruct kernels) + a library
    GEMM -- the reference's own rule above 50 (8-bit: 24) rows."""
    lib = _lib.lib()
    m, k = a.shape
    n, groups = qweight.shape[1], scales.shape[0]
    has_perm = g_idx is not None and g_idx.numel() > 0
    if use_exllama and 0 < m <= 64 and lib.aphro_gptq_gemm_bits_supported(min(m, 32), n, k, groups, bit):
        x = a[:, g_idx.long()] if has_perm else a           # act-order: gather once (q_gemm.cu:219-226)
        if x.stride(1) != 1 or x.stride(0) % 8 != 0 or x.data_ptr() % 16 != 0:
            x = x.contiguous()
        out = torch.empty((m, n), dtype=a.dtype, device=a.device)
        for m0 in range(0, m, 32):
            rows = min(32, m - m0)
            check(lib.aphro_gptq_gemm_bits(x[m0:].data_ptr(), x.stride(0), qweight.data_ptr(), qzeros.data_ptr(),
                                           scales.data_ptr(), out[m0:].data_ptr(), rows, n, k, groups, bit, _dt(a),
                                           _stream()), "gptq_gemm")
        return out
    w = gptq_dequant(qweight, qzeros, scales, g_idx, use_exllama, bit)
    if use_exllama and has_perm:
        a = a[:, g_idx.long()]
    _library_fallback("gptq_gemm", f"{bit}-bit weights at M={m}: gptq_dequant + matmul")
    return torch.matmul(a, w)


def gptq_marlin_repack(b_q_weight: torch.Tensor, perm: torch.Tensor, size_k: int,
                       size_n: int, num_bits: int) -> torch.Tensor:
    """Marlin-role load-time prepack into the CDNA4 K-packed layout (same
    shape as the input, [K/8, N]); perm = argsort(g_idx) or empty.  num_bits = 8 (uint8b128, marlin_utils.py:28-45): the
    checkpoint's sequential [K/4, N] words are what the 8-bit kernels read (csrc/wnx_gemm.hip) -- act-order rows made
    sequential, nothing else moves."""
    _require_cuda(b_q_weight)
    if num_bits == 8:
        out = b_q_weight.clone()
        if perm is not None and perm.numel() > 0:
            gptq_shuffle(out, perm, 8)
        return out
    out = torch.empty_like(b_q_weight)
    p = perm.to(torch.int32) if (perm is not None and perm.numel() > 0) else None
    check(_lib.lib().aphro_gptq_repack(b_q_weight.data_ptr(), _ptr(p),
                                       out.data_ptr(), size_k, size_n, num_bits,
                                       _stream()), "gptq_marlin_repack")
    return out


# --------------------------------------------------------------------------
# AWQ
# --------------------------------------------------------------------------
_ZP8 = {}


def _uint4b8_zeros(device: torch.device, groups: int, n: int) -> torch.Tensor:
    """Packed zero points of the symmetric uint4b8 type (8 in every nibble), made once per shape: no allocation
    or fill launch inside the op (it runs under HIP-graph capture)."""
    key = (device.type, device.index, groups, n)
    z = _ZP8.get(key)
    if z is None:
        z = torch.full((groups, n // 8), 0x88888888 - (1 << 32), dtype=torch.int64, device=device).to(torch.int32)
        _ZP8[key] = z
    return z


def gptq_marlin_gemm(a: torch.Tensor, b_q_weight: torch.Tensor, b_scales: torch.Tensor,
                     b_zeros: torch.Tensor, g_idx: torch.Tensor, perm: torch.Tensor,
                     workspace: Optional[torch.Tensor], b_q_type, size_m: int, size_n: int,
                     size_k: int, is_k_full: bool = True, has_zp: bool = False,
                     use_fp32_reduce: bool = False, is_zp_float: bool = False) -> torch.Tensor:
    """W4A16 "Marlin role" (torch_bindings.cpp:195-201) on the CDNA4 layout:
    b_q_weight = gptq_marlin_repack / awq_marlin_repack output ([K/8, N] K-packed, exllama
    nibble order; act-order rows already sorted), b_scales [G, N], b_zeros int32 [G, N/8]
    plain column order (awq_repack_zeros output) when has_zp, ignored for the symmetric
    uint4b8 type (zero point 8).  perm = argsort(g_idx) or empty; g_idx / workspace /
    is_k_full / use_fp32_reduce are accepted for signature parity (the kernel always
    reduces in fp32 and needs no lock workspace)."""
    if is_zp_float:
        raise RuntimeError("gptq_marlin_gemm: float zero points are not supported")
    bits = getattr(b_q_type, "size_bits", 4)
    if bits == 8:
        # uint8b128 (quantization/utils/marlin_utils.py:28-45): b_q_weight = gptq_marlin_repack(..., 8) -- sequential [K/4, N]
        # words; the symmetric zero point 128 in GPTQ's stored-minus-one convention is 127 per byte; perm gathers the
        # activation columns (gptq_gemm's exllama form)
        if has_zp or getattr(b_q_type, "bias", 128) != 128:
            raise RuntimeError("gptq_marlin_gemm: the 8-bit type served is uint8b128 (symmetric, no zero points)")
        x = a.reshape(-1, a.shape[-1])
        if x.shape[0] != size_m or x.shape[1] != size_k or b_q_weight.shape != (size_k // 4, size_n):
            raise RuntimeError("gptq_marlin_gemm: shape mismatch")
        key = (a.device.type, a.device.index, b_scales.shape[0], size_n, 8)
        zp = _ZP8.get(key)
        if zp is None:
            zp = torch.full((b_scales.shape[0], size_n // 4), 0x7f7f7f7f, dtype=torch.int32, device=a.device)
            _ZP8[key] = zp
        p = perm if perm is not None and perm.numel() > 0 else torch.empty(0, dtype=torch.int32, device=a.device)
        return gptq_gemm(x, b_q_weight, zp, b_scales, p, True, 8)
    if bits != 4:
        raise RuntimeError("gptq_marlin_gemm on MI355X serves uint4 / uint4b8 and uint8b128 weights")
    x = a.reshape(-1, a.shape[-1])
    if x.shape[0] != size_m or x.shape[1] != size_k or b_q_weight.shape != (size_k // 8, size_n):
        raise RuntimeError("gptq_marlin_gemm: shape mismatch")
    groups = b_scales.shape[0]
    if has_zp:
        zp = b_zeros
    else:
        zp = _uint4b8_zeros(a.device, groups, size_n)
    p = perm if perm is not None and perm.numel() > 0 else None
    return wna16_gemm(x, b_q_weight, zp, b_scales, p, 0)


def awq_dequantize(qweight: torch.Tensor, scales: torch.Tensor,
                   zeros: torch.Tensor, split_k_iters: int = 0, thx: int = 0,
                   thy: int = 0) -> torch.Tensor:
    _require_cuda(qweight, scales, zeros)
    k, n = qweight.shape[0], qweight.shape[1] * 8
    out = torch.empty((k, n), dtype=scales.dtype, device=qweight.device)
    check(_lib.lib().aphro_awq_dequantize(
        qweight.data_ptr(), scales.data_ptr(), zeros.data_ptr(), out.data_ptr(),
        k, n, scales.shape[0], _dt(scales), _stream()), "awq_dequantize")
    return out


AWQ_REPACK_MIN_M = 256    # awq_gemm rows from which the per-call nibble transpose + prefill-sized kernel pays


def awq_gemm(input: torch.Tensor, qweight: torch.Tensor, qzeros: torch.Tensor,
             scales: torch.Tensor, split_k_iters: int) -> torch.Tensor:
    """NOTE the reference wrapper's parameter names are swapped relative to what
    callers pass (SURVEY 8b gotcha): positionally it is
    (in_feats, kernel, scaling_factors, zeros, split_k_iters) --
    awq.py:165-166 calls awq_gemm(x, qweight, scales, qzeros, pack_factor).
    So here ``qzeros`` receives the scales and ``scales`` the zeros."""
    scaling_factors, zeros = qzeros, scales
    _require_cuda(input, qweight, scaling_factors, zeros)
    lib = _lib.lib()
    m, k = input.shape
    n = qweight.shape[1] * 8
    groups = scaling_factors.shape[0]
    if input.stride(1) != 1:
        input = input.contiguous()
    if m >= AWQ_REPACK_MIN_M and wna16_large_ok(m, n, k, groups) and not switch("APHRO_WNA16_NO_LARGE"):
        # prefill-sized M on checkpoint-layout AWQ tensors: transpose the nibbles once per call (2 x the weight bytes,
        # a few % of the GEMM at this M) and run the MFMA-bound kernel -- the reference dequantises + matmuls above 256
        # tokens (awq.py:160-164).  Load-time repack (AWQConfig(prepack=True)) skips this step entirely.
        return _wna16_large(input, awq_marlin_repack(qweight, k, n, 4), awq_repack_zeros(zeros, n), scaling_factors,
                            None, 0)
    out = torch.empty((m, n), dtype=input.dtype, device=input.device)
    nbytes = lib.aphro_awq_gemm_workspace_bytes(min(m, 64), n, k, groups)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=input.device)
    for m0 in range(0, m, 64):
        rows = min(64, m - m0)
        check(lib.aphro_awq_gemm(
            input[m0:].data_ptr(), qweight.data_ptr(), scaling_factors.data_ptr(),
            zeros.data_ptr(), out[m0:].data_ptr(), ws.data_ptr(), ws.numel(),
            rows, n, k, groups, input.stride(0), _dt(input), _stream()),
            "awq_gemm")
    return out


def awq_marlin_repack(b_q_weight: torch.Tensor, size_k: int, size_n: int,
                      num_bits: int) -> torch.Tensor:
    """AWQ [K, N/8] -> CDNA4 K-packed [K/8, N] (load time)."""
    _require_cuda(b_q_weight)
    if num_bits != 4:
        raise RuntimeError("only 4-bit AWQ is implemented")
    out = torch.empty((size_k // 8, size_n), dtype=torch.int32,
                      device=b_q_weight.device)
    check(_lib.lib().aphro_awq_repack(b_q_weight.data_ptr(), out.data_ptr(),
                                      size_k, size_n, _stream()), "awq_repack")
    return out


def awq_repack_zeros(qzeros: torch.Tensor, size_n: int) -> torch.Tensor:
    out = torch.empty_like(qzeros)
    check(_lib.lib().aphro_awq_repack_zeros(qzeros.data_ptr(), out.data_ptr(),
                                            qzeros.shape[0], size_n, _stream()),
          "awq_repack_zeros")
    return out


# -- decode fast path: fragment-major activations + fused glue --------------------
def wna16_ksplit(m: int, n: int, k: int, groups: int) -> int:
    """fp32 split-K slabs the fast kernel produces for this shape (0: not served)."""
    return _lib.lib().aphro_wna16_ksplit(m, n, k, groups)


def wna16_pack_a(a: torch.Tensor, perm: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[M,K] f16/bf16 -> fragment-major f16 buffer (see include/aphrodite_mi355x.h)."""
    _require_cuda(a)
    m, k = a.shape
    lib = _lib.lib()
    out = torch.empty(lib.aphro_wna16_packed_a_bytes(m, k) // 2, dtype=torch.float16, device=a.device)
    if a.stride(1) != 1:
        a = a.contiguous()
    check(lib.aphro_wna16_pack_a(a.data_ptr(), _ptr(perm), out.data_ptr(), m, k, a.stride(0),
                                 _dt(a), _stream()), "wna16_pack_a")
    return out


def wna16_gemm_packed(a_packed: torch.Tensor, m: int, k: int, qweight: torch.Tensor,
                      qzeros: torch.Tensor, scales: torch.Tensor, zero_offset: int,
                      partials: bool = False):
    """GEMM on packed activations.  partials=False -> [M,N] tensor in scales.dtype;
    partials=True -> (fp32 slabs [S,M,N], S) left for a fused consumer."""
    lib = _lib.lib()
    n = qweight.shape[1]
    groups = scales.shape[0]
    ks = lib.aphro_wna16_ksplit(m, n, k, groups)
    if ks <= 0:
        raise RuntimeError(f"wna16_gemm_packed: shape M={m} N={n} K={k} not served by the fast kernel")
    dt = _dt(scales)
    if partials:
        slabs = torch.empty((ks, m, n), dtype=torch.float32, device=qweight.device)
        check(lib.aphro_wna16_gemm_packed(a_packed.data_ptr(), qweight.data_ptr(), qzeros.data_ptr(),
                                          scales.data_ptr(), None, slabs.data_ptr(), slabs.numel() * 4,
                                          m, n, k, groups, zero_offset, dt, _stream()), "wna16_gemm_packed")
        return slabs, ks
    out = torch.empty((m, n), dtype=scales.dtype, device=qweight.device)
    slabs = torch.empty((ks, m, n), dtype=torch.float32, device=qweight.device) if ks > 1 else None
    check(lib.aphro_wna16_gemm_packed(a_packed.data_ptr(), qweight.data_ptr(), qzeros.data_ptr(),
                                      scales.data_ptr(), out.data_ptr(), _ptr(slabs),
                                      slabs.numel() * 4 if slabs is not None else 0, m, n, k, groups,
                                      zero_offset, dt, _stream()), "wna16_gemm_packed")
    return out


def interleave_gate_up(qweight: torch.Tensor, qzeros: torch.Tensor, scales: torch.Tensor):
    """Load-time column permutation of a merged [gate | up] K-packed int4 tensor set so that
    column 2j = gate_j and 2j+1 = up_j (the layout wna16_gemm_silu_pack consumes).
    qweight int32 [K/8, 2I] (any row order: the shuffle is per word along K), qzeros int32
    [G, 2I/8] (8 nibbles per word along N), scales [G, 2I]."""
    n = qweight.shape[1]
    half = n // 2
    idx = torch.arange(n, device=qweight.device)
    src = torch.where(idx % 2 == 0, idx // 2, half + idx // 2)
    return _permute_columns(qweight, qzeros, scales, src)


def deinterleave_gate_up(qweight: torch.Tensor, qzeros: torch.Tensor, scales: torch.Tensor):
    """The inverse of interleave_gate_up: [gate | up] column order back from (gate_j, up_j) pairs."""
    n = qweight.shape[1]
    half = n // 2
    idx = torch.arange(n, device=qweight.device)
    src = torch.where(idx < half, 2 * idx, 2 * (idx - half) + 1)
    return _permute_columns(qweight, qzeros, scales, src)


def _permute_columns(qweight: torch.Tensor, qzeros: torch.Tensor, scales: torch.Tensor, src: torch.Tensor):
    """Column j of the result = column src[j] of a K-packed int4 tensor set (zero points: 8 nibbles per word along N)."""
    n = qweight.shape[1]
    qw = qweight[:, src].contiguous()
    sc = scales[:, src].contiguous()
    shifts = torch.arange(0, 32, 4, device=qzeros.device, dtype=torch.int32)
    z = ((qzeros.unsqueeze(-1) >> shifts) & 0xF).reshape(qzeros.shape[0], n)[:, src]
    z = z.reshape(qzeros.shape[0], n // 8, 8).to(torch.int64)
    packed = torch.zeros(qzeros.shape[0], n // 8, dtype=torch.int64, device=qzeros.device)
    for i in range(8):
        packed |= z[:, :, i] << (4 * i)
    packed = torch.where(packed >= 2 ** 31, packed - 2 ** 32, packed).to(torch.int32)
    return qw, packed.contiguous(), sc


def wna16_gemm_silu_pack(a_packed: torch.Tensor, m: int, k: int, qweight: torch.Tensor,
                         qzeros: torch.Tensor, scales: torch.Tensor, zero_offset: int) -> torch.Tensor:
    """gate_up GEMM (interleaved columns, see interleave_gate_up) + SiluAndMul + pack:
    returns the fragment-major f16 activations [M, N/2] for the down projection."""
    lib = _lib.lib()
    n = qweight.shape[1]
    groups = scales.shape[0]
    out = torch.empty(lib.aphro_wna16_packed_a_bytes(m, n // 2) // 2, dtype=torch.float16,
                      device=qweight.device)
    check(lib.aphro_wna16_gemm_silu_pack(a_packed.data_ptr(), qweight.data_ptr(), qzeros.data_ptr(),
                                         scales.data_ptr(), out.data_ptr(), m, n, k, groups, zero_offset,
                                         _dt(scales), _stream()), "wna16_gemm_silu_pack")
    return out


def wna16_gemm_mid_ksplit(m: int, n: int, k: int, groups: int) -> int:
    """K slices the 33..64-row kernel would use on the decode fast path; 0: shape not served."""
    return int(_lib.lib().aphro_wna16_gemm_mid_ksplit(m, n, k, groups))


def _mid_packed_call(lib, strip_m: int, *args):
    """aphro_wna16_gemm_mid_packed, or its strip-major form when ``strip_m`` > 0 (args: ..., dtype, stream)."""
    if strip_m:
        return lib.aphro_wna16_gemm_mid_packed_strip(*args[:-1], strip_m, args[-1])
    return lib.aphro_wna16_gemm_mid_packed(*args)


def wna16_gemm_mid_packed(a_packed: torch.Tensor, m: int, k: int, qweight: torch.Tensor, qzeros: torch.Tensor,
                          scales: torch.Tensor, zero_offset: int, partials: bool = False, strip_m: int = 0):
    """wna16_gemm_packed for 33..64 rows on the one-pass MFMA kernel (csrc/wna16_gemm_mid.hip) -- same packed
    activations in, same conventions out: partials=True -> (fp32 slabs [S, M, N], S) for a fused consumer.
    ``strip_m`` > 0: ``qweight`` is the strip-major copy laid out for that M class (the only resident one)."""
    lib = _lib.lib()
    n = qweight.shape[1]
    groups = scales.shape[0]
    ks = lib.aphro_wna16_gemm_mid_ksplit(m, n, k, groups)
    if ks <= 0:
        raise RuntimeError(f"wna16_gemm_mid_packed: shape M={m} N={n} K={k} not served")
    if partials:
        slabs = torch.empty((ks, m, n), dtype=torch.float32, device=qweight.device)
        check(_mid_packed_call(lib, strip_m, a_packed.data_ptr(), qweight.data_ptr(), qzeros.data_ptr(), scales.data_ptr(),
                               None, slabs.data_ptr(), slabs.numel() * 4, None, m, n, k, groups,
                               zero_offset, _dt(scales), _stream()), "wna16_gemm_mid_packed")
        return slabs, ks
    if ks > 1:
        slabs, _ = wna16_gemm_mid_packed(a_packed, m, k, qweight, qzeros, scales, zero_offset, partials=True, strip_m=strip_m)
        return slabs.sum(0).to(scales.dtype)
    out = torch.empty((m, n), dtype=scales.dtype, device=qweight.device)
    check(_mid_packed_call(lib, strip_m, a_packed.data_ptr(), qweight.data_ptr(), qzeros.data_ptr(), scales.data_ptr(),
                           out.data_ptr(), None, 0, None, m, n, k, groups, zero_offset, _dt(scales),
                           _stream()), "wna16_gemm_mid_packed")
    return out


def wna16_gemm_mid_silu_pack(a_packed: torch.Tensor, m: int, k: int, qweight: torch.Tensor, qzeros: torch.Tensor,
                             scales: torch.Tensor, zero_offset: int, strip_m: int = 0) -> torch.Tensor:
    """wna16_gemm_silu_pack for 33..64 rows: gate_up GEMM (interleaved columns) + SiluAndMul + pack in one launch."""
    lib = _lib.lib()
    n = qweight.shape[1]
    out = torch.empty(lib.aphro_wna16_packed_a_bytes(m, n // 2) // 2, dtype=torch.float16, device=qweight.device)
    check(_mid_packed_call(lib, strip_m, a_packed.data_ptr(), qweight.data_ptr(), qzeros.data_ptr(), scales.data_ptr(),
                           None, None, 0, out.data_ptr(), m, n, k, scales.shape[0], zero_offset,
                           _dt(scales), _stream()), "wna16_gemm_mid_silu_pack")
    return out


def wna16_resident_ksplit(m: int, n: int, k: int, groups: int) -> int:
    """K slices of the resident-activation decode kernel (csrc/wna16_gemm_resident.hip) for this shape; 0: not served."""
    return int(_lib.lib().aphro_wna16_resident_ksplit(m, n, k, groups))


def wna16_strip_relayout(qweight: torch.Tensor, m: int, groups: int) -> torch.Tensor:
    """Load-time: [K/8, N] exllama-ordered words -> the strip-major order the resident kernel streams (a permutation)."""
    _require_cuda(qweight)
    out = torch.empty_like(qweight)
    check(_lib.lib().aphro_wna16_strip_relayout(qweight.data_ptr(), out.data_ptr(), m, qweight.shape[1],
                                                qweight.shape[0] * 8, groups, _stream()), "wna16_strip_relayout")
    return out


def wna16_gemm_resident(a_packed: torch.Tensor, m: int, k: int, qweight: torch.Tensor, qzeros: torch.Tensor,
                        scales: torch.Tensor, zero_offset: int, mode: str = "out", strip_layout: bool = False):
    """Decode GEMM (M <= 32) on packed activations with the resident kernel.  mode "out": [M, N] tensor (one K slice
    only); "slabs": (fp32 [S, M, N], S) for a fused consumer; "silu": SiluAndMul + pack epilogue over interleaved
    gate / up columns -> fragment-major f16 [M, N/2]."""
    lib = _lib.lib()
    n = scales.shape[1]
    groups = scales.shape[0]
    ks = lib.aphro_wna16_resident_ksplit(m, n, k, groups)
    if ks <= 0:
        raise RuntimeError(f"wna16_gemm_resident: shape M={m} N={n} K={k} not served")
    dev = qweight.device
    c = slabs = act = None
    if mode == "silu":
        act = torch.empty(lib.aphro_wna16_packed_a_bytes(m, n // 2) // 2, dtype=torch.float16, device=dev)
    elif mode == "slabs":
        slabs = torch.empty((ks, m, n), dtype=torch.float32, device=dev)
    else:
        c = torch.empty((m, n), dtype=scales.dtype, device=dev)
    check(lib.aphro_wna16_gemm_resident(a_packed.data_ptr(), qweight.data_ptr(), qzeros.data_ptr(), scales.data_ptr(),
                                        _ptr(c), _ptr(slabs), slabs.numel() * 4 if slabs is not None else 0, _ptr(act),
                                        m, n, k, groups, zero_offset, _dt(scales), 1 if strip_layout else 0, _stream()),
          "wna16_gemm_resident")
    if mode == "silu":
        return act
    if mode == "slabs":
        return slabs, ks
    return c


def wna16_gemm_rowmajor_supported(m: int, n: int, k: int, groups: int, dtype: torch.dtype) -> bool:
    return dtype == torch.float16 and bool(_lib.lib().aphro_wna16_gemm_rowmajor_supported(m, n, k, groups, _DT[dtype]))


def wna16_gemm_rowmajor(a: torch.Tensor, qweight: torch.Tensor, qzeros: torch.Tensor, scales: torch.Tensor,
                        zero_offset: int, strip_layout: bool = False) -> torch.Tensor:
    """The op-level decode GEMM (M <= 32, f16) in ONE launch: row-major ``a`` read in place, K slices reduced inside the
    kernel (csrc/wna16_gemm_resident.hip).  ``gptq_gemm`` / ``awq_gemm`` take this path by themselves; this entry exists
    for the strip-major weight copy and the tests."""
    lib = _lib.lib()
    m, k = a.shape
    n, groups = scales.shape[1], scales.shape[0]
    if a.stride(1) != 1 or a.stride(0) % 8 != 0 or a.data_ptr() % 16 != 0:
        a = a.contiguous()
    out = torch.empty((m, n), dtype=a.dtype, device=a.device)
    nbytes = lib.aphro_wna16_workspace_bytes(m, n, k)
    ws = _workspace(a.device, nbytes)
    check(lib.aphro_wna16_gemm_rowmajor(a.data_ptr(), a.stride(0), qweight.data_ptr(), qzeros.data_ptr(), scales.data_ptr(),
                                        out.data_ptr(), ws.data_ptr(), ws.numel(), m, n, k, groups, zero_offset, _dt(a),
                                        1 if strip_layout else 0, _stream()), "wna16_gemm_rowmajor")
    return out


STRIP_COPY_MIN_BYTES = 32 << 20


def wna16_decode_strip_copy(qweight: torch.Tensor, scales: torch.Tensor) -> Optional[torch.Tensor]:
    """Load-time: the strip-major copy of a [K/8, N] int4 matrix for the one-launch decode GEMM, or None where it does
    not pay.  Measured (tools/op_gemm_bench.py, profiles/r3_op_gemm.txt): 21.2 -> 18.8 us on the 4096 x 28672 gate_up
    matrix at 32 rows (two column passes per workgroup), nothing on the one-pass shapes (down / qkv / o) -- so only
    matrices of >= 32 MiB get one (it doubles their footprint: the [K/8, N] original still serves M > 32).
    APHRODITE_MI355X_NO_STRIP_COPY=1 turns it off."""
    if switch("APHRODITE_MI355X_NO_STRIP_COPY") == "1" or scales.dtype != torch.float16:
        return None
    k, n, groups = qweight.shape[0] * 8, qweight.shape[1], scales.shape[0]
    if qweight.numel() * 4 < STRIP_COPY_MIN_BYTES or not wna16_gemm_rowmajor_supported(32, n, k, groups, torch.float16):
        return None
    return wna16_strip_relayout(qweight, 32, groups)


def wna16_decode_linear(x: torch.Tensor, qweight: torch.Tensor, qzeros: torch.Tensor, scales: torch.Tensor,
                        zero_offset: int, strip: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """The strip-major one-launch GEMM when the call is a decode batch it serves, else None (caller: the generic op)."""
    if strip is None or x.shape[0] > 32 or x.dtype != torch.float16 or x.shape[0] == 0:
        return None
    if not wna16_gemm_rowmajor_supported(x.shape[0], scales.shape[1], x.shape[1], scales.shape[0], x.dtype):
        return None
    return wna16_gemm_rowmajor(x, strip, qzeros, scales, zero_offset, strip_layout=True)


def fused_add_rms_norm_pack(x: Optional[torch.Tensor], slabs: Optional[torch.Tensor],
                            residual: Optional[torch.Tensor], has_residual: bool,
                            weight: torch.Tensor, epsilon: float, pack: bool = True,
                            want_out: bool = False):
    """[slab reduce] + fused_add_rms_norm + [pack]; returns (packed or None, out or None)."""
    lib = _lib.lib()
    if slabs is not None:
        nslab, tokens, hidden = slabs.shape
        dev = slabs.device
    else:
        tokens, hidden = x.shape
        nslab, dev = 0, x.device
        assert x.is_contiguous()
    packed = torch.empty(lib.aphro_wna16_packed_a_bytes(tokens, hidden) // 2, dtype=torch.float16,
                         device=dev) if pack else None
    out = torch.empty((tokens, hidden), dtype=weight.dtype, device=dev) if want_out else None
    check(lib.aphro_fused_add_rms_norm_pack(_ptr(x), _ptr(slabs), nslab, _ptr(residual),
                                            1 if has_residual else 0, weight.data_ptr(), float(epsilon),
                                            _ptr(packed), _ptr(out), tokens, hidden, _dt(weight), _stream()),
          "fused_add_rms_norm_pack")
    return packed, out


def fused_add_rms_norm_router(x: Optional[torch.Tensor], slabs: Optional[torch.Tensor], residual: Optional[torch.Tensor],
                              has_residual: bool, weight: torch.Tensor, epsilon: float, router_weight: torch.Tensor):
    """[slab reduce] + fused_add_rms_norm + the router's logits of a sparse-MLP layer (the replicated ``gate`` linear of
    MixtralMoE, mixtral.py:60-110) in one launch; returns (out [tokens, hidden], router_logits [tokens, E]), E <= 16."""
    lib = _lib.lib()
    if slabs is not None:
        nslab, tokens, hidden = slabs.shape
        dev = slabs.device
    else:
        tokens, hidden = x.shape
        nslab, dev = 0, x.device
        assert x.is_contiguous()
    e = router_weight.shape[0]
    if router_weight.shape[1] != hidden or not router_weight.is_contiguous() or router_weight.dtype != weight.dtype:
        raise RuntimeError("fused_add_rms_norm_router: router_weight must be a contiguous [E, hidden] tensor of the norm's dtype")
    out = torch.empty((tokens, hidden), dtype=weight.dtype, device=dev)
    logits = torch.empty((tokens, e), dtype=weight.dtype, device=dev)
    check(lib.aphro_fused_add_rms_norm_router(_ptr(x), _ptr(slabs), nslab, _ptr(residual), 1 if has_residual else 0,
                                              weight.data_ptr(), float(epsilon), out.data_ptr(), router_weight.data_ptr(),
                                              logits.data_ptr(), e, tokens, hidden, _dt(weight), _stream()),
          "fused_add_rms_norm_router")
    return out, logits


def fused_add_rms_norm_pack_combine(slabs: torch.Tensor, inv_pos: torch.Tensor, topk_weights: torch.Tensor,
                                    residual: Optional[torch.Tensor], has_residual: bool, weight: torch.Tensor,
                                    epsilon: float, pack: bool = True, want_out: bool = False):
    """moe_combine + fused_add_rms_norm (+ pack) in one launch: the norm that follows a sparse MLP reads the second expert
    GEMM's slabs through inv_pos / topk_weights itself.  Returns (packed or None, out or None)."""
    lib = _lib.lib()
    nslab, m_pad, hidden = slabs.shape
    tokens, topk = topk_weights.shape
    dev = slabs.device
    packed = torch.empty(lib.aphro_wna16_packed_a_bytes(tokens, hidden) // 2, dtype=torch.float16, device=dev) if pack else None
    out = torch.empty((tokens, hidden), dtype=weight.dtype, device=dev) if want_out else None
    check(lib.aphro_fused_add_rms_norm_pack_combine(slabs.data_ptr(), nslab, m_pad, inv_pos.data_ptr(),
                                                    topk_weights.data_ptr(), topk, _ptr(residual), 1 if has_residual else 0,
                                                    weight.data_ptr(), float(epsilon), _ptr(packed), _ptr(out), tokens,
                                                    hidden, _dt(weight), _stream()), "fused_add_rms_norm_pack_combine")
    return packed, out


def silu_and_mul_pack(x: Optional[torch.Tensor], slabs: Optional[torch.Tensor] = None,
                      dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """SiluAndMul + pack of [gate | up] rows; with ``slabs`` (fp32 [S, tokens, 2 d], ``dtype`` = the activation dtype) the
    split-K reduce of the gate_up GEMM rides in the same launch (same bits as reduce, then this op)."""
    lib = _lib.lib()
    if slabs is not None:
        nslab, tokens, d2 = slabs.shape
        d = d2 // 2
        assert slabs.is_contiguous() and slabs.dtype == torch.float32 and dtype in _DT
        packed = torch.empty(lib.aphro_wna16_packed_a_bytes(tokens, d) // 2, dtype=torch.float16, device=slabs.device)
        check(lib.aphro_silu_and_mul_pack_slabs(slabs.data_ptr(), nslab, packed.data_ptr(), None, tokens, d, _DT[dtype],
                                                _stream()), "silu_and_mul_pack_slabs")
        return packed
    tokens, d2 = x.shape
    d = d2 // 2
    assert x.is_contiguous()
    packed = torch.empty(lib.aphro_wna16_packed_a_bytes(tokens, d) // 2, dtype=torch.float16, device=x.device)
    check(lib.aphro_silu_and_mul_pack(x.data_ptr(), packed.data_ptr(), None, tokens, d, _dt(x), _stream()),
          "silu_and_mul_pack")
    return packed


def rope_cache(qkv: Optional[torch.Tensor], slabs: Optional[torch.Tensor], positions: torch.Tensor,
               cos_sin_cache: torch.Tensor, is_neox: bool, key_cache: torch.Tensor,
               value_cache: torch.Tensor, slot_mapping: torch.Tensor, num_heads: int,
               num_kv_heads: int, head_size: int, kv_cache_dtype: str, k_scale: float,
               v_scale: float) -> torch.Tensor:
    """[slab reduce] + rotary_embedding(q, k) + reshape_and_cache(k, v); returns q [T, Hq*hd]."""
    lib = _lib.lib()
    if slabs is not None:
        nslab, tokens, _ = slabs.shape
        dev, stride = slabs.device, 0
    else:
        tokens = qkv.shape[0]
        nslab, dev, stride = 0, qkv.device, qkv.stride(0)
    dtype = cos_sin_cache.dtype
    q_out = torch.empty((tokens, num_heads * head_size), dtype=dtype, device=dev)
    if positions is not None and positions.dtype != torch.int64:
        positions = positions.long()
    check(lib.aphro_rope_cache(_ptr(qkv), stride, _ptr(slabs), nslab, _ptr(positions),
                               cos_sin_cache.data_ptr(), cos_sin_cache.shape[1], 1 if is_neox else 0,
                               q_out.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(),
                               slot_mapping.data_ptr(), tokens, num_heads, num_kv_heads, head_size,
                               key_cache.shape[3], key_cache.shape[4], _dt(cos_sin_cache),
                               _kv(kv_cache_dtype), float(k_scale), float(v_scale), _stream()), "rope_cache")
    return q_out


def paged_attention_packed(query: torch.Tensor, key_cache: torch.Tensor, value_cache: torch.Tensor,
                           num_kv_heads: int, scale: float, block_tables: torch.Tensor,
                           seq_lens: torch.Tensor, block_size: int, max_seq_len: int,
                           alibi_slopes: Optional[torch.Tensor], kv_cache_dtype: str, k_scale: float,
                           v_scale: float, want_out: bool = False):
    """Single-launch decode attention writing its output fragment-major for o_proj."""
    lib = _lib.lib()
    num_seqs, num_heads, head_size = query.shape
    packed = torch.empty(lib.aphro_wna16_packed_a_bytes(num_seqs, num_heads * head_size) // 2,
                         dtype=torch.float16, device=query.device)
    out = torch.empty((num_seqs, num_heads, head_size), dtype=query.dtype, device=query.device) \
        if want_out else None
    check(lib.aphro_paged_attention_packed(
        _ptr(out), packed.data_ptr(), query.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(),
        num_seqs, num_heads, num_kv_heads, head_size, float(scale), block_tables.data_ptr(),
        seq_lens.data_ptr(), block_tables.stride(0), block_size, int(max_seq_len), _ptr(alibi_slopes),
        query.stride(0), key_cache.stride(0), key_cache.stride(1), _dt(query), _kv(kv_cache_dtype),
        float(k_scale), float(v_scale), _stream()), "paged_attention_packed")
    return packed, out


def paged_attention_rope_packed(qkv_slabs: torch.Tensor, positions: torch.Tensor,
                                cos_sin_cache: torch.Tensor, slot_mapping: torch.Tensor,
                                key_cache: torch.Tensor, value_cache: torch.Tensor, num_heads: int,
                                num_kv_heads: int, scale: float, block_tables: torch.Tensor,
                                seq_lens: torch.Tensor, block_size: int, max_seq_len: int,
                                alibi_slopes: Optional[torch.Tensor], kv_cache_dtype: str, k_scale: float,
                                v_scale: float, want_out: bool = False):
    """[qkv slab reduce] + rotary_embedding + reshape_and_cache + decode attention + pack in one
    launch (head_size 128, NeoX, v1 form).  Mutates key_cache / value_cache.  positions=None:
    cos_sin_cache is the per-sequence gather cos_sin[positions] (done once per step)."""
    lib = _lib.lib()
    nslab, num_seqs, ntot = qkv_slabs.shape
    head_size = ntot // (num_heads + 2 * num_kv_heads)
    dtype = cos_sin_cache.dtype
    packed = torch.empty(lib.aphro_wna16_packed_a_bytes(num_seqs, num_heads * head_size) // 2,
                         dtype=torch.float16, device=qkv_slabs.device)
    out = torch.empty((num_seqs, num_heads, head_size), dtype=dtype, device=qkv_slabs.device) \
        if want_out else None
    if positions is not None and positions.dtype != torch.int64:
        positions = positions.long()
    check(lib.aphro_paged_attention_rope_packed(
        _ptr(out), packed.data_ptr(), qkv_slabs.data_ptr(), nslab, _ptr(positions),
        cos_sin_cache.data_ptr(), slot_mapping.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(),
        num_seqs, num_heads, num_kv_heads, head_size, float(scale), block_tables.data_ptr(),
        seq_lens.data_ptr(), block_tables.stride(0), block_size, int(max_seq_len), _ptr(alibi_slopes),
        key_cache.stride(0), key_cache.stride(1), _dt(cos_sin_cache), _kv(kv_cache_dtype),
        float(k_scale), float(v_scale), _stream()), "paged_attention_rope_packed")
    return packed, out


# --------------------------------------------------------------------------
# random sampling
# --------------------------------------------------------------------------
def sample_top_k_top_p(logits: torch.Tensor, temperature: Optional[torch.Tensor] = None,
                       top_k: Optional[torch.Tensor] = None, top_p: Optional[torch.Tensor] = None,
                       q: Optional[torch.Tensor] = None, seeds: Optional[torch.Tensor] = None,
                       out: Optional[torch.Tensor] = None, min_p: Optional[torch.Tensor] = None,
                       logprobs_out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """temperature -> top-k -> top-p -> min-p -> softmax -> argmax(probs / q) in one launch (sampler.py:256-262,
    865-891, 894-908, 1273-1292).  logits [B, V] f16 / bf16 / f32 (rows may be strided); per-row fp32
    temperature / int32 top_k / fp32 top_p or None; q: Exp(1) draws [B, V] fp32 (``torch.empty_like
    (...).exponential_()`` like the reference), or None with int64 ``seeds`` [B] to draw in the kernel.
    ``logprobs_out`` (fp32 [B], optional) receives log_softmax of the row after temperature and masks -- what
    the reference calls logprobs (sampler.py:545) -- at the sampled token.  Returns int64 [B]."""
    _require_cuda(logits)
    if logits.dim() != 2 or logits.stride(1) != 1:
        raise RuntimeError("sample_top_k_top_p: logits must be [rows, vocab] with unit column stride")
    rows, vocab = logits.shape
    dev = logits.device

    def prep(t_, dtype):
        if t_ is None:
            return None
        if t_.dtype != dtype or not t_.is_contiguous() or t_.device != dev:
            t_ = t_.to(device=dev, dtype=dtype).contiguous()
        if t_.numel() != rows:
            raise RuntimeError("sample_top_k_top_p: per-row parameters must have one entry per row")
        return t_
    temperature, top_k, top_p = prep(temperature, torch.float32), prep(top_k, torch.int32), prep(top_p, torch.float32)
    min_p = prep(min_p, torch.float32)
    seeds = prep(seeds, torch.int64)
    if q is not None and (q.dtype != torch.float32 or q.shape != logits.shape or q.stride(1) != 1):
        raise RuntimeError("sample_top_k_top_p: q must be float32 [rows, vocab]")
    if q is None and seeds is None:
        raise RuntimeError("sample_top_k_top_p: pass the Exp(1) noise q or per-row seeds")
    if out is None:
        out = torch.empty(rows, dtype=torch.int64, device=dev)
    if logprobs_out is not None and (logprobs_out.dtype != torch.float32 or logprobs_out.numel() != rows
                                     or not logprobs_out.is_contiguous() or logprobs_out.device != dev):
        raise RuntimeError("sample_top_k_top_p: logprobs_out must be a contiguous float32 [rows] tensor on the device")
    check(_lib.lib().aphro_sample_top_k_top_p(
        out.data_ptr(), logits.data_ptr(), logits.stride(0), _ptr(temperature), _ptr(top_k), _ptr(top_p), _ptr(min_p),
        _ptr(q),
        q.stride(0) if q is not None else 0, _ptr(seeds), _ptr(logprobs_out), rows, vocab, _dt(logits), _stream()),
        "sample_top_k_top_p")
    return out


# --------------------------------------------------------------------------
# FP8 (W8A8, per-token dynamic) decode fast path
# --------------------------------------------------------------------------
def fp8_gemm_ksplit(m: int, n: int, k: int) -> int:
    """Split-K factor scaled_mm_fp8_slabs uses for this shape (<= 0: shape not served)."""
    return int(_lib.lib().aphro_fp8_gemm_ksplit(m, n, k))


def scaled_mm_fp8_slabs(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """a fp8 [M,K], b fp8 [K,N] column-major (weight.t()); returns the raw fp32 accumulators as
    split-K slabs [S, M, N] -- the consumer applies a_scale * (b_scale * sum)."""
    lib = _lib.lib()
    m, k = a.shape
    n = b.shape[1]
    if b.stride(0) != 1 or b.stride(1) != k:
        raise RuntimeError("b must be column-major [K,N] (weight.t())")
    ks = lib.aphro_fp8_gemm_ksplit(m, n, k)
    if ks <= 0:
        raise RuntimeError(f"scaled_mm_fp8_slabs: shape M={m} N={n} K={k} not served")
    slabs = torch.empty((ks, m, n), dtype=torch.float32, device=a.device)
    check(lib.aphro_scaled_mm_fp8_slabs(a.data_ptr(), b.data_ptr(), slabs.data_ptr(), slabs.numel() * 4,
                                        m, n, k, _stream()), "scaled_mm_fp8_slabs")
    return slabs


def fp8_gemm_resident_ksplit(m: int, n: int, k: int) -> int:
    """K slices of the resident FP8 decode GEMM's plan for [m, k] x [n, k]^T (0: shape not served; m <= 32)."""
    return int(_lib.lib().aphro_fp8_gemm_resident_ksplit(m, n, k))


def fp8_strip_relayout(weight: torch.Tensor, m: int = 32) -> torch.Tensor:
    """Load time: the [N, K] e4m3 checkpoint tensor -> the strip-major copy ops.fp8_gemm_resident streams (every wave's
    1 KiB pieces in the order it reads them; csrc/fp8_gemm_resident.hip).  Same bytes, permuted."""
    check_fp8_buffer(weight, "fp8_strip_relayout")
    if weight.dim() != 2 or not weight.is_contiguous():
        raise RuntimeError("fp8_strip_relayout: weight must be a contiguous [N, K] tensor")
    n, k = weight.shape
    out = torch.empty_like(weight)
    check(_lib.lib().aphro_fp8_strip_relayout(weight.data_ptr(), out.data_ptr(), m, n, k, _stream()), "fp8_strip_relayout")
    return out


def fp8_gemm_resident(a: torch.Tensor, w_strip: torch.Tensor, scale_a: Optional[torch.Tensor] = None,
                      scale_b: Optional[torch.Tensor] = None, out_dtype: Optional[torch.dtype] = None,
                      bias: Optional[torch.Tensor] = None, slabs: bool = False) -> torch.Tensor:
    """FP8 W8A8 decode GEMM (<= 32 rows) on the strip-major copy of a [N, K] weight: ``slabs=True`` returns the raw fp32
    accumulators [ksplit, M, N] (the role of scaled_mm_fp8_slabs), else ``scale_a * (scale_b * acc) (+ bias)`` in
    ``out_dtype`` (the role of cutlass_scaled_mm; plans with one K slice only)."""
    m, k = a.shape
    n = w_strip.shape[0]
    if w_strip.shape[1] != k or not a.is_contiguous() or not w_strip.is_contiguous():
        raise RuntimeError("fp8_gemm_resident: a [M, K] and the strip-major copy of a [N, K] weight expected")
    check_fp8_buffer(a, "fp8_gemm_resident")
    check_fp8_buffer(w_strip, "fp8_gemm_resident")
    lib = _lib.lib()
    ks = lib.aphro_fp8_gemm_resident_ksplit(m, n, k)
    if ks <= 0:
        raise RuntimeError(f"fp8_gemm_resident: shape M={m} N={n} K={k} not served")
    if slabs:
        out = torch.empty((ks, m, n), dtype=torch.float32, device=a.device)
        check(lib.aphro_fp8_gemm_resident(a.data_ptr(), k, w_strip.data_ptr(), None, None, None, None, out.data_ptr(),
                                          out.numel() * 4, m, n, k, 0, 0, _dt_of(torch.float16), _stream()), "fp8_gemm_resident")
        return out
    if ks != 1:
        raise RuntimeError(f"fp8_gemm_resident: M={m} N={n} K={k} is K-sliced ({ks}): slabs only")
    sa = scale_a.to(torch.float32).contiguous()
    sb = scale_b.to(torch.float32).contiguous()
    if bias is not None and (bias.dtype != out_dtype or bias.numel() != n):
        raise RuntimeError("fp8_gemm_resident: bias must be [N] in the output dtype")
    out = torch.empty((m, n), dtype=out_dtype, device=a.device)
    check(lib.aphro_fp8_gemm_resident(a.data_ptr(), k, w_strip.data_ptr(), sa.data_ptr(), sb.data_ptr(), _ptr(bias),
                                      out.data_ptr(), None, 0, m, n, k, 1 if sa.numel() > 1 else 0,
                                      1 if sb.numel() > 1 else 0, _dt_of(out_dtype), _stream()), "fp8_gemm_resident")
    return out


def fp8_strip_relayout_interleaved(weight: torch.Tensor, m: int = 32) -> torch.Tensor:
    """fp8_strip_relayout for a [gate; up] matrix whose SiluAndMul runs in the GEMM epilogue (ops.fp8_gemm_resident_silu):
    strip column 2 j = gate row j, 2 j + 1 = up row N / 2 + j."""
    check_fp8_buffer(weight, "fp8_strip_relayout_interleaved")
    if weight.dim() != 2 or not weight.is_contiguous():
        raise RuntimeError("fp8_strip_relayout_interleaved: weight must be a contiguous [N, K] tensor")
    n, k = weight.shape
    out = torch.empty_like(weight)
    check(_lib.lib().aphro_fp8_strip_relayout_interleaved(weight.data_ptr(), out.data_ptr(), m, n, k, _stream()),
          "fp8_strip_relayout_interleaved")
    return out


def fp8_gemm_resident_strips(m: int, n: int, k: int) -> int:
    """Column strips of the resident plan for (M, N, K) = absmax partials per row of ops.fp8_gemm_resident_silu; 0: not served."""
    return int(_lib.lib().aphro_fp8_gemm_resident_strips(m, n, k))


def fp8_gemm_resident_silu_supported(m: int, n: int, k: int) -> bool:
    return n % 2 == 0 and _lib.lib().aphro_fp8_gemm_resident_ksplit(m, n, k) == 1


def aq_pairs_numel(m: int, k: int) -> int:
    """16-bit elements of the pair-major activation buffer for [m, k] (k % 64 == 0): [k / 64][ceil(m / 16)][2][64 lanes][8]."""
    return (k // 64) * ((m + 15) // 16) * 1024


def aq_pairs_index(m: int, k: int, device) -> torch.Tensor:
    """int64 [m, k]: the offset of element (row, column) in the pair-major buffer (csrc/common.h aq_pair_offset) -- the layout
    the fused producers write for ops.fp8_gemm_resident_aq (every 16-byte load of the GEMM lane-linear).  Tests / debugging."""
    row = torch.arange(m, device=device).view(-1, 1)
    col = torch.arange(k, device=device).view(1, -1)
    mtiles = (m + 15) // 16
    return ((((col >> 6) * mtiles + (row >> 4)) * 2 + ((col & 15) >> 3)) * 64 + ((col & 63) >> 4) * 16 + (row & 15)) * 8 + (col & 7)


def fp8_gemm_resident_silu(a: torch.Tensor, w_strip_il: torch.Tensor, scale_a: torch.Tensor, scale_b: torch.Tensor,
                           out_dtype: torch.dtype, static_out_scale: Optional[torch.Tensor] = None, act_pairs: bool = False):
    """gate_up W8A8 GEMM + SiluAndMul in one launch on the interleaved strip-major copy (<= 32 rows).  Dynamic scheme:
    returns (act [M, N / 2] in ``out_dtype``, absmax partials fp32 [M, strips]) -- the consumer
    (ops.fp8_gemm_resident_aq) turns the partials into dynamic_per_token_scaled_fp8_quant's scale and quantises on load.
    ``static_out_scale`` (fp32 [1], the down projection's input_scale): returns the e4m3 [M, N / 2]
    static_scaled_fp8_quant would produce from act.  Bits of cutlass_scaled_mm -> silu_and_mul [-> the quantiser]."""
    m, k = a.shape
    n = w_strip_il.shape[0]
    if w_strip_il.shape[1] != k or not a.is_contiguous() or not w_strip_il.is_contiguous():
        raise RuntimeError("fp8_gemm_resident_silu: a [M, K] and the interleaved strip-major copy of a [N, K] weight expected")
    check_fp8_buffer(a, "fp8_gemm_resident_silu")
    check_fp8_buffer(w_strip_il, "fp8_gemm_resident_silu")
    lib = _lib.lib()
    strips = lib.aphro_fp8_gemm_resident_strips(m, n, k)
    if strips <= 0 or lib.aphro_fp8_gemm_resident_ksplit(m, n, k) != 1:
        raise RuntimeError(f"fp8_gemm_resident_silu: shape M={m} N={n} K={k} not served with one K slice")
    sa = scale_a.to(torch.float32).contiguous()
    sb = scale_b.to(torch.float32).contiguous()
    if static_out_scale is not None:
        _check_static_scale(static_out_scale, a.device)
        q8 = torch.empty((m, n // 2), dtype=FP8_DTYPE, device=a.device)
        check(lib.aphro_fp8_gemm_resident_silu(a.data_ptr(), k, w_strip_il.data_ptr(), sa.data_ptr(), sb.data_ptr(), None, 0, None,
                                               q8.data_ptr(), static_out_scale.data_ptr(), m, n, k, 1 if sa.numel() > 1 else 0,
                                               1 if sb.numel() > 1 else 0, _dt_of(out_dtype), _stream()), "fp8_gemm_resident_silu")
        return q8
    # act_pairs: the activation in the pair-major layout (a flat buffer, aq_pairs_index) for ops.fp8_gemm_resident_aq(a_pairs=...)
    if act_pairs and (n // 2) % 64 != 0:
        raise RuntimeError("fp8_gemm_resident_silu: the pair-major activation needs N / 2 % 64 == 0")
    act = torch.empty((aq_pairs_numel(m, n // 2), ) if act_pairs else (m, n // 2), dtype=out_dtype, device=a.device)
    absmax = torch.empty((m, strips), dtype=torch.float32, device=a.device)
    check(lib.aphro_fp8_gemm_resident_silu(a.data_ptr(), k, w_strip_il.data_ptr(), sa.data_ptr(), sb.data_ptr(), act.data_ptr(),
                                           1 if act_pairs else 0, absmax.data_ptr(), None, None, m, n, k, 1 if sa.numel() > 1 else 0,
                                           1 if sb.numel() > 1 else 0, _dt_of(out_dtype), _stream()), "fp8_gemm_resident_silu")
    return act, absmax


def fp8_gemm_resident_aq_supported(m: int, n: int, k: int, np_: int) -> bool:
    return 4 <= np_ <= 256 and np_ % 4 == 0 and _lib.lib().aphro_fp8_gemm_resident_ksplit(m, n, k) > 0


def fp8_gemm_resident_aq(a16: torch.Tensor, absmax: torch.Tensor, w_strip: torch.Tensor, slabs: bool = True,
                         scale_b: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None,
                         a_pairs: bool = False):
    """The resident W8A8 GEMM fed with the PRODUCER's 16-bit activations ``a16`` [M, K] and its absmax partials
    ``absmax`` fp32 [M, np]: per-token scales and e4m3 operands are made inside the launch -- the bits of
    ops.scaled_fp8_quant(a16, None, use_per_token_if_dynamic=True) -> ops.fp8_gemm_resident.  ``a_pairs``: a16 is the flat
    pair-major buffer a fused producer wrote (aq_pairs_index).  Returns (slabs fp32 [ksplit, M, N] or out [M, N] in a16's
    dtype, scales fp32 [M, 1])."""
    n, k = w_strip.shape
    if absmax.dtype != torch.float32 or absmax.dim() != 2 or not absmax.is_contiguous():
        raise RuntimeError("fp8_gemm_resident_aq: absmax partials must be a contiguous fp32 [M, np] tensor")
    m = absmax.shape[0]
    if a16.dtype not in (torch.float16, torch.bfloat16) or not a16.is_contiguous() or not w_strip.is_contiguous() or \
            (a16.numel() != aq_pairs_numel(m, k) if a_pairs else tuple(a16.shape) != (m, k)):
        raise RuntimeError("fp8_gemm_resident_aq: a16 [M, K] f16 / bf16 (or its pair-major buffer) and the strip-major copy of a [N, K] weight expected")
    check_fp8_buffer(w_strip, "fp8_gemm_resident_aq")
    lib = _lib.lib()
    ks = lib.aphro_fp8_gemm_resident_ksplit(m, n, k)
    if ks <= 0:
        raise RuntimeError(f"fp8_gemm_resident_aq: shape M={m} N={n} K={k} not served")
    sc = torch.empty((m, 1), dtype=torch.float32, device=a16.device)
    ap = 1 if a_pairs else 0
    if slabs:
        out = torch.empty((ks, m, n), dtype=torch.float32, device=a16.device)
        check(lib.aphro_fp8_gemm_resident_aq(a16.data_ptr(), k, ap, absmax.data_ptr(), absmax.shape[1], w_strip.data_ptr(),
                                             sc.data_ptr(), None, None, None, out.data_ptr(), out.numel() * 4, m, n, k, 0,
                                             _dt(a16), _stream()), "fp8_gemm_resident_aq")
        return out, sc
    if ks != 1:
        raise RuntimeError(f"fp8_gemm_resident_aq: M={m} N={n} K={k} is K-sliced ({ks}): slabs only")
    sb = scale_b.to(torch.float32).contiguous()
    out = torch.empty((m, n), dtype=a16.dtype, device=a16.device)
    check(lib.aphro_fp8_gemm_resident_aq(a16.data_ptr(), k, ap, absmax.data_ptr(), absmax.shape[1], w_strip.data_ptr(),
                                         sc.data_ptr(), sb.data_ptr(), _ptr(bias), out.data_ptr(), None, 0, m, n, k,
                                         1 if sb.numel() > 1 else 0, _dt(a16), _stream()), "fp8_gemm_resident_aq")
    return out, sc


def fp8_quant_rows_aq(x: torch.Tensor, absmax: torch.Tensor):
    """The quantiser of ops.fp8_gemm_resident_aq on its own (same device code): (q e4m3 [M, K], scales fp32 [M, 1]) from the
    16-bit rows x [M, K] and absmax partials fp32 [M, np] -- for the parity tests against x / scale."""
    m, k = x.shape
    if not x.is_contiguous() or absmax.dtype != torch.float32 or not absmax.is_contiguous() or absmax.shape[0] != m:
        raise RuntimeError("fp8_quant_rows_aq: contiguous x [M, K] and fp32 absmax [M, np] expected")
    q = torch.empty((m, k), dtype=FP8_DTYPE, device=x.device)
    sc = torch.empty((m, 1), dtype=torch.float32, device=x.device)
    check(_lib.lib().aphro_fp8_quant_rows_aq(x.data_ptr(), absmax.data_ptr(), absmax.shape[1], q.data_ptr(), sc.data_ptr(), m, k,
                                             _dt(x), _stream()), "fp8_quant_rows_aq")
    return q, sc


def fp8_gemm_silu_quant_supported(m: int, n: int, k: int) -> bool:
    return bool(_lib.lib().aphro_fp8_gemm_stream_silu_supported(m, n, k))


def fp8_gemm_silu_quant(a: torch.Tensor, b: torch.Tensor, scale_a: torch.Tensor, scale_b: torch.Tensor,
                        static_scale: torch.Tensor, dtype: torch.dtype, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """gate_up GEMM + SiluAndMul + static fp8 quant in one launch: a fp8 [M, K], b fp8 [K, N] column-major
    (gate_up weight.t()); returns e4m3 [M, N / 2] = static_scaled_fp8_quant(silu_and_mul(cutlass_scaled_mm(a, b, ...)),
    static_scale), bit for bit, with ``dtype`` the dtype the two intermediate tensors would have had."""
    m, k = a.shape
    n = b.shape[1]
    if b.stride(0) != 1 or b.stride(1) != k or not a.is_contiguous():
        raise RuntimeError("fp8_gemm_silu_quant: a [M, K] contiguous and b column-major [K, N] (weight.t()) expected")
    check_fp8_buffer(a, "fp8_gemm_silu_quant")
    check_fp8_buffer(b, "fp8_gemm_silu_quant")
    _check_static_scale(static_scale, a.device)
    q = torch.empty((m, n // 2), dtype=FP8_DTYPE, device=a.device)
    sa = scale_a.to(torch.float32).contiguous()
    sb = scale_b.to(torch.float32).contiguous()
    if bias is not None and (bias.dtype != dtype or bias.numel() != n):
        raise RuntimeError("fp8_gemm_silu_quant: bias must be [N] in the activation dtype")
    check(_lib.lib().aphro_fp8_gemm_stream_silu_quant(
        a.data_ptr(), k, b.data_ptr(), sa.data_ptr(), sb.data_ptr(), _ptr(bias), q.data_ptr(), static_scale.data_ptr(),
        m, n, k, 1 if sa.numel() > 1 else 0, 1 if sb.numel() > 1 else 0, _dt_of(dtype), _stream()), "fp8_gemm_silu_quant")
    return q


def _check_static_scale(static_scale: Optional[torch.Tensor], dev) -> None:
    if static_scale is not None and (static_scale.dtype != torch.float32 or static_scale.numel() != 1
                                     or static_scale.device.type != torch.device(dev).type):
        raise RuntimeError("static_scale must be one float32 on the activations' device")


def fused_add_rms_norm_quant_fp8(x: Optional[torch.Tensor], slabs: Optional[torch.Tensor],
                                 slab_a_scales: Optional[torch.Tensor], slab_b_scales: Optional[torch.Tensor],
                                 residual: torch.Tensor, has_residual: bool, weight: torch.Tensor,
                                 epsilon: float, want_out: bool = False,
                                 static_scale: Optional[torch.Tensor] = None):
    """[slab reduce + dequant] + fused_add_rms_norm + per-token fp8 quant (``static_scale``: the layer's per-tensor
    input_scale, fp32 [1] -- quantises as static_scaled_fp8_quant and returns that scale for every token).
    Returns (q fp8 [M, H], scales fp32 [M, 1], out or None)."""
    lib = _lib.lib()
    if slabs is not None:
        nslab, tokens, hidden = slabs.shape
        dev = slabs.device
    else:
        tokens, hidden = x.shape
        nslab, dev = 0, x.device
        assert x.is_contiguous()
    q = torch.empty((tokens, hidden), dtype=FP8_DTYPE, device=dev)
    sc = torch.empty((tokens, 1), dtype=torch.float32, device=dev)
    out = torch.empty((tokens, hidden), dtype=weight.dtype, device=dev) if want_out else None
    a_tok = 1 if slab_a_scales is not None and slab_a_scales.numel() > 1 else 0
    b_ch = 1 if slab_b_scales is not None and slab_b_scales.numel() > 1 else 0
    _check_static_scale(static_scale, dev)
    check(lib.aphro_fused_add_rms_norm_quant_fp8_static(
        _ptr(x), _ptr(slabs), nslab, _ptr(slab_a_scales), _ptr(slab_b_scales), a_tok, b_ch, residual.data_ptr(),
        1 if has_residual else 0, weight.data_ptr(), float(epsilon), q.data_ptr(), sc.data_ptr(), _ptr(out),
        tokens, hidden, _dt(weight), _ptr(static_scale), _stream()), "fused_add_rms_norm_quant_fp8")
    return q, sc, out


def silu_and_mul_quant_fp8(x: torch.Tensor, want_out: bool = False, static_scale: Optional[torch.Tensor] = None):
    """silu_and_mul + per-token fp8 quant over x [M, 2d] (``static_scale``: as in fused_add_rms_norm_quant_fp8);
    returns (q [M, d], scales [M, 1], out or None)."""
    lib = _lib.lib()
    tokens, d2 = x.shape
    d = d2 // 2
    assert x.is_contiguous()
    q = torch.empty((tokens, d), dtype=FP8_DTYPE, device=x.device)
    sc = torch.empty((tokens, 1), dtype=torch.float32, device=x.device)
    out = torch.empty((tokens, d), dtype=x.dtype, device=x.device) if want_out else None
    _check_static_scale(static_scale, x.device)
    check(lib.aphro_silu_and_mul_quant_fp8_static(x.data_ptr(), q.data_ptr(), sc.data_ptr(), _ptr(out), tokens, d,
                                                  _dt(x), _ptr(static_scale), _stream()), "silu_and_mul_quant_fp8")
    return q, sc, out


def paged_attention_rope_scaled(qkv_slabs: torch.Tensor, slab_row_scale: Optional[torch.Tensor],
                                slab_col_scale: torch.Tensor, positions: Optional[torch.Tensor],
                                cos_sin_cache: torch.Tensor, slot_mapping: torch.Tensor, key_cache: torch.Tensor,
                                value_cache: torch.Tensor, num_heads: int, num_kv_heads: int, scale: float,
                                block_tables: torch.Tensor, seq_lens: torch.Tensor, block_size: int,
                                max_seq_len: int, alibi_slopes: Optional[torch.Tensor], kv_cache_dtype: str,
                                k_scale: float, v_scale: float,
                                out_q8_scale: Optional[torch.Tensor] = None, want_out: bool = True,
                                want_absmax: bool = False, out_pairs: bool = False):
    """paged_attention_rope_packed over the raw slabs of an FP8 qkv projection (dequantised on the
    fly); returns the attention output [S, Hq, hd] row-major in the activation dtype.
    ``out_q8_scale`` (fp32 [1]: the static input_scale of an FP8 o_proj): returns (out or None, out_q8) with out_q8 the
    e4m3 [S, Hq * hd] static_scaled_fp8_quant would produce from out -- written by the attention launch itself."""
    lib = _lib.lib()
    nslab, num_seqs, ntot = qkv_slabs.shape
    head_size = ntot // (num_heads + 2 * num_kv_heads)
    if positions is not None and positions.dtype != torch.int64:
        positions = positions.long()
    if out_q8_scale is not None:
        _check_static_scale(out_q8_scale, qkv_slabs.device)
        out = torch.empty((num_seqs, num_heads, head_size), dtype=cos_sin_cache.dtype, device=qkv_slabs.device) \
            if want_out else None
        out_q8 = torch.empty((num_seqs, num_heads * head_size), dtype=FP8_DTYPE, device=qkv_slabs.device)
        check(lib.aphro_paged_attention_rope_scaled_q8(
            _ptr(out), out_q8.data_ptr(), out_q8_scale.data_ptr(), qkv_slabs.data_ptr(), nslab, _ptr(slab_row_scale),
            slab_col_scale.data_ptr(), _ptr(positions), cos_sin_cache.data_ptr(), slot_mapping.data_ptr(),
            key_cache.data_ptr(), value_cache.data_ptr(), num_seqs, num_heads, num_kv_heads, head_size, float(scale),
            block_tables.data_ptr(), seq_lens.data_ptr(), block_tables.stride(0), block_size, int(max_seq_len),
            _ptr(alibi_slopes), key_cache.stride(0), key_cache.stride(1), _dt(cos_sin_cache), _kv(kv_cache_dtype),
            float(k_scale), float(v_scale), _stream()), "paged_attention_rope_scaled_q8")
        return out, out_q8
    if want_absmax:
        # dynamic per-token scheme: (out, absmax partials [S, Hkv]) -- the o_proj GEMM (ops.fp8_gemm_resident_aq) makes the
        # per-token scale from the partials and quantises `out` on load; out_pairs: `out` is the flat pair-major buffer of
        # [S, Hq * hd] that GEMM reads lane-linearly (aq_pairs_index) instead of the row-major [S, Hq, hd]
        if out_pairs:
            out = torch.empty((aq_pairs_numel(num_seqs, num_heads * head_size), ), dtype=cos_sin_cache.dtype, device=qkv_slabs.device)
        else:
            out = torch.empty((num_seqs, num_heads, head_size), dtype=cos_sin_cache.dtype, device=qkv_slabs.device)
        absmax = torch.empty((num_seqs, num_kv_heads), dtype=torch.float32, device=qkv_slabs.device)
        check(lib.aphro_paged_attention_rope_scaled_absmax(
            None if out_pairs else out.data_ptr(), out.data_ptr() if out_pairs else None, absmax.data_ptr(), qkv_slabs.data_ptr(),
            nslab, _ptr(slab_row_scale),
            slab_col_scale.data_ptr(), _ptr(positions), cos_sin_cache.data_ptr(), slot_mapping.data_ptr(),
            key_cache.data_ptr(), value_cache.data_ptr(), num_seqs, num_heads, num_kv_heads, head_size, float(scale),
            block_tables.data_ptr(), seq_lens.data_ptr(), block_tables.stride(0), block_size, int(max_seq_len),
            _ptr(alibi_slopes), key_cache.stride(0), key_cache.stride(1), _dt(cos_sin_cache), _kv(kv_cache_dtype),
            float(k_scale), float(v_scale), _stream()), "paged_attention_rope_scaled_absmax")
        return out, absmax
    out = torch.empty((num_seqs, num_heads, head_size), dtype=cos_sin_cache.dtype, device=qkv_slabs.device)
    check(lib.aphro_paged_attention_rope_packed_scaled(
        out.data_ptr(), None, qkv_slabs.data_ptr(), nslab, _ptr(slab_row_scale), slab_col_scale.data_ptr(),
        _ptr(positions), cos_sin_cache.data_ptr(), slot_mapping.data_ptr(), key_cache.data_ptr(),
        value_cache.data_ptr(), num_seqs, num_heads, num_kv_heads, head_size, float(scale),
        block_tables.data_ptr(), seq_lens.data_ptr(), block_tables.stride(0), block_size, int(max_seq_len),
        _ptr(alibi_slopes), key_cache.stride(0), key_cache.stride(1), _dt(cos_sin_cache), _kv(kv_cache_dtype),
        float(k_scale), float(v_scale), _stream()), "paged_attention_rope_packed_scaled")
    return out


# --------------------------------------------------------------------------
# per-step bookkeeping
# --------------------------------------------------------------------------
def advance_step_flashattn(num_seqs: int, num_queries: int, block_size: int, input_tokens: torch.Tensor,
                           sampled_token_ids: torch.Tensor, input_positions: torch.Tensor,
                           seq_lens: torch.Tensor, slot_mapping: torch.Tensor,
                           block_tables: torch.Tensor) -> None:
    """_custom_ops.py advance_step_flashattn (kernels/torch_bindings.cpp:77-82)."""
    _require_cuda(input_tokens, sampled_token_ids, input_positions, seq_lens, slot_mapping, block_tables)
    for name, t_, dt in (("input_tokens", input_tokens, torch.int64), ("sampled_token_ids", sampled_token_ids, torch.int64),
                         ("input_positions", input_positions, torch.int64), ("seq_lens", seq_lens, torch.int32),
                         ("slot_mapping", slot_mapping, torch.int64), ("block_tables", block_tables, torch.int32)):
        if t_.dtype != dt or not t_.is_contiguous():
            raise RuntimeError(f"tensor: name = {name}, shape = {tuple(t_.shape)} is_cont = {t_.is_contiguous()}, "
                               f"type = {t_.dtype} is not as expected: type = {dt}")
    check(_lib.lib().aphro_advance_step_flashattn(
        num_seqs, num_queries, block_size, input_tokens.data_ptr(), sampled_token_ids.data_ptr(),
        input_positions.data_ptr(), seq_lens.data_ptr(), slot_mapping.data_ptr(), block_tables.data_ptr(),
        block_tables.stride(0), _stream()), "advance_step_flashattn")


def argmax_rows(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Greedy sampling: int64 argmax over the last dim of a 2-D logits tensor (lowest index on ties)."""
    _require_cuda(x)
    if x.dim() != 2 or x.stride(1) != 1:
        raise RuntimeError("argmax_rows: expected a 2-D tensor with contiguous rows")
    if out is None:
        out = torch.empty(x.shape[0], dtype=torch.int64, device=x.device)
    check(_lib.lib().aphro_argmax_rows(out.data_ptr(), x.data_ptr(), x.shape[0], x.shape[1], x.stride(0),
                                       _dt(x), _stream()), "argmax_rows")
    return out


def lm_head_argmax_supported(m: int, k: int, v: int, ldw: int, dtype: torch.dtype) -> bool:
    return dtype in (torch.float16, torch.bfloat16) and \
        bool(_lib.lib().aphro_lm_head_argmax_supported(m, k, v, ldw, _DT[dtype]))


def lm_head_argmax(hidden: torch.Tensor, weight: torch.Tensor, vocab_size: Optional[int] = None,
                   out: Optional[torch.Tensor] = None, logits_out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Greedy decode of M <= 32 rows in one launch: int64 argmax over ``hidden @ weight[:vocab_size].T`` rounded to the
    activation dtype (LogitsProcessor._get_logits + Sampler._greedy_sample, logits_processor.py:78-96 / sampler.py);
    ``logits_out`` [M, >= vocab_size] also receives the logits."""
    _require_cuda(hidden, weight)
    if hidden.dim() != 2 or weight.dim() != 2 or hidden.dtype != weight.dtype or hidden.shape[1] != weight.shape[1]:
        raise RuntimeError("lm_head_argmax: hidden [M, K] and weight [V, K] of one dtype expected")
    if hidden.stride(1) != 1 or weight.stride(1) != 1:
        raise RuntimeError("lm_head_argmax: rows must be contiguous")
    v = weight.shape[0] if vocab_size is None else int(vocab_size)
    if v > weight.shape[0]:
        raise RuntimeError("lm_head_argmax: vocab_size exceeds the weight rows")
    m, k = hidden.shape
    if out is None:
        out = torch.empty(m, dtype=torch.int64, device=hidden.device)
    if logits_out is not None and (logits_out.dtype != hidden.dtype or logits_out.stride(1) != 1 or logits_out.shape[0] < m):
        raise RuntimeError("lm_head_argmax: logits_out must be [M, >= V] of the activation dtype with contiguous rows")
    check(_lib.lib().aphro_lm_head_argmax(hidden.data_ptr(), hidden.stride(0), weight.data_ptr(), weight.stride(0),
                                          _ptr(logits_out), logits_out.stride(0) if logits_out is not None else 0,
                                          out.data_ptr(), m, k, v, _dt(hidden), _stream()), "lm_head_argmax")
    return out


# --------------------------------------------------------------------------
# mixture of experts
# --------------------------------------------------------------------------
def fp8_moe_gemm(a_q: torch.Tensor, w: torch.Tensor, a_scale: torch.Tensor, w_scales: torch.Tensor,
                 topk_weights: Optional[torch.Tensor], sorted_ids: torch.Tensor, expert_ids: torch.Tensor,
                 num_post_pad: torch.Tensor, out: torch.Tensor, top_k_div: int) -> None:
    """One grouped FP8 W8A8 GEMM of an MoE layer (fused_moe_kernel with use_fp8_w8a8, fused_moe.py:20-170): for every
    valid slot of the expert-sorted list, out[slot] = T(((a_q[slot // top_k_div] . w[expert]^T) * w_routed[slot]) *
    a_scale * w_scales[expert])."""
    _require_cuda(a_q, w, out)
    check_fp8_buffer(a_q, "fp8_moe_gemm")
    check_fp8_buffer(w, "fp8_moe_gemm")
    e, n, k = w.shape
    if a_q.dim() != 2 or a_q.shape[1] != k or not a_q.is_contiguous() or not w.is_contiguous() or not out.is_contiguous():
        raise RuntimeError("fp8_moe_gemm: contiguous a_q [rows, K], w [E, N, K], out [slots, N] expected")
    if out.shape[1] != n or w_scales.numel() != e or a_scale.numel() != 1:
        raise RuntimeError("fp8_moe_gemm: shape mismatch")
    tw = None
    if topk_weights is not None:
        tw = topk_weights.to(torch.float32).contiguous()
        if tw.numel() != out.shape[0]:
            raise RuntimeError("fp8_moe_gemm: one routed weight per slot expected")
    check(_lib.lib().aphro_fp8_moe_gemm(a_q.data_ptr(), w.data_ptr(), a_scale.to(torch.float32).data_ptr(),
                                        w_scales.to(torch.float32).contiguous().data_ptr(), _ptr(tw), sorted_ids.data_ptr(),
                                        expert_ids.data_ptr(), num_post_pad.data_ptr(), out.data_ptr(), out.shape[0], n, k,
                                        expert_ids.numel(), int(top_k_div), _dt(out), _stream()), "fp8_moe_gemm")


def topk_softmax(topk_weights: torch.Tensor, topk_ids: torch.Tensor,
                 token_expert_indicies: Optional[torch.Tensor], gating_output: torch.Tensor) -> None:
    """_custom_ops.py topk_softmax -> _moe_C::topk_softmax (kernels/moe/torch_bindings.cpp:11-14)."""
    _require_cuda(topk_weights, topk_ids, gating_output)
    if gating_output.dtype != torch.float32 or not gating_output.is_contiguous():
        raise RuntimeError("topk_softmax: gating_output must be contiguous float32")
    if topk_ids.dtype != torch.int32 or topk_weights.dtype != torch.float32:
        raise RuntimeError("topk_softmax: topk_weights must be float32 and topk_ids int32")
    t_, e = gating_output.shape
    check(_lib.lib().aphro_topk_softmax(topk_weights.data_ptr(), topk_ids.data_ptr(),
                                        _ptr(token_expert_indicies), gating_output.data_ptr(), t_, e,
                                        topk_ids.shape[1], _stream()), "topk_softmax")


def moe_align_block_size(topk_ids: torch.Tensor, num_experts: int, block_size: int,
                         sorted_token_ids: torch.Tensor, experts_ids: torch.Tensor,
                         num_tokens_post_pad: torch.Tensor, inv_pos: Optional[torch.Tensor] = None) -> None:
    """_custom_ops.py moe_align_block_size (kernels/torch_bindings.cpp:394-399); inv_pos is an
    optional extra output of ours (position of every (token, k) slot in the sorted order)."""
    _require_cuda(topk_ids, sorted_token_ids, experts_ids, num_tokens_post_pad)
    if topk_ids.dtype != torch.int32 or not topk_ids.is_contiguous():
        raise RuntimeError("moe_align_block_size: topk_ids must be contiguous int32")
    numel = topk_ids.numel()
    need = numel + num_experts * (block_size - 1)
    if sorted_token_ids.numel() < need or experts_ids.numel() < (need + block_size - 1) // block_size:
        raise RuntimeError("moe_align_block_size: output tensors too small")
    check(_lib.lib().aphro_moe_align_block_size(topk_ids.data_ptr(), num_experts, block_size,
                                                sorted_token_ids.data_ptr(), experts_ids.data_ptr(),
                                                num_tokens_post_pad.data_ptr(), _ptr(inv_pos), numel,
                                                _stream()), "moe_align_block_size")


def moe_route_align(gating_output: torch.Tensor, topk: int, renormalize: bool, num_experts: int, block_size: int,
                    want_inverse: bool = False):
    """fused_topk + moe_align_block_size in one launch (decode-sized batches; csrc/moe.hip moe_route_align_kernel):
    returns (topk_weights fp32 [T, k], topk_ids int32 [T, k], sorted_ids, expert_ids, num_tokens_post_pad, inv or None) --
    bit-identical to topk_softmax (+ the fp32 renormalisation) followed by moe_align_block_size."""
    _require_cuda(gating_output)
    t_, e = gating_output.shape
    if e != num_experts or gating_output.stride(1) != 1:
        raise RuntimeError("moe_route_align: gating_output must be [tokens, num_experts] with contiguous rows")
    dev = gating_output.device
    numel = t_ * topk
    max_padded = numel + num_experts * (block_size - 1)
    topk_weights = torch.empty((t_, topk), dtype=torch.float32, device=dev)
    topk_ids = torch.empty((t_, topk), dtype=torch.int32, device=dev)
    sorted_ids = torch.empty((max_padded, ), dtype=torch.int32, device=dev)
    expert_ids = torch.empty(((max_padded + block_size - 1) // block_size, ), dtype=torch.int32, device=dev)
    post_pad = torch.empty((1, ), dtype=torch.int32, device=dev)
    inv = torch.empty(numel, dtype=torch.int32, device=dev) if want_inverse else None
    check(_lib.lib().aphro_moe_route_align(topk_weights.data_ptr(), topk_ids.data_ptr(), gating_output.data_ptr(),
                                           gating_output.stride(0), sorted_ids.data_ptr(), expert_ids.data_ptr(),
                                           post_pad.data_ptr(), _ptr(inv), t_, num_experts, topk, 1 if renormalize else 0,
                                           block_size, _dt(gating_output), _stream()), "moe_route_align")
    return topk_weights, topk_ids, sorted_ids, expert_ids, post_pad, inv


MOE_ROUTE_ALIGN_MAX_SLOTS = 8192


def moe_route_gather_supported(num_tokens: int, num_experts: int, topk: int, block_size: int, k: int) -> bool:
    return bool(_lib.lib().aphro_moe_route_gather_supported(num_tokens, num_experts, topk, block_size, k))


def moe_route_gather(hidden_states: torch.Tensor, gating_output: torch.Tensor, topk: int, renormalize: bool, num_experts: int,
                     block_size: int):
    """moe_route_align + moe_gather_pack in ONE launch (csrc/moe.hip moe_route_gather_kernel; decode-sized batches, <= 16
    experts): returns (topk_weights, topk_ids, sorted_ids, expert_ids, num_tokens_post_pad, inv, packed, m_pad) -- bit for bit
    what the two ops return."""
    _require_cuda(hidden_states, gating_output)
    t_, e = gating_output.shape
    if e != num_experts or gating_output.stride(1) != 1 or gating_output.dtype != hidden_states.dtype \
            or hidden_states.stride(1) != 1 or hidden_states.shape[0] != t_:
        raise RuntimeError("moe_route_gather: [tokens, E] logits and [tokens, K] activations of one 16-bit dtype, contiguous rows")
    dev = gating_output.device
    k = hidden_states.shape[1]
    numel = t_ * topk
    max_padded = numel + num_experts * (block_size - 1)
    m_pad = (max_padded + 15) // 16 * 16
    topk_weights = torch.empty((t_, topk), dtype=torch.float32, device=dev)
    topk_ids = torch.empty((t_, topk), dtype=torch.int32, device=dev)
    sorted_ids = torch.empty((max_padded, ), dtype=torch.int32, device=dev)
    expert_ids = torch.empty(((max_padded + block_size - 1) // block_size, ), dtype=torch.int32, device=dev)
    post_pad = torch.empty((1, ), dtype=torch.int32, device=dev)
    inv = torch.empty(numel, dtype=torch.int32, device=dev)
    lib = _lib.lib()
    packed = torch.empty(lib.aphro_wna16_packed_a_bytes(m_pad, k) // 2, dtype=torch.float16, device=dev)
    check(lib.aphro_moe_route_gather(topk_weights.data_ptr(), topk_ids.data_ptr(), gating_output.data_ptr(),
                                     gating_output.stride(0), sorted_ids.data_ptr(), expert_ids.data_ptr(), post_pad.data_ptr(),
                                     inv.data_ptr(), t_, num_experts, topk, 1 if renormalize else 0, block_size,
                                     hidden_states.data_ptr(), hidden_states.stride(0), packed.data_ptr(), m_pad, k,
                                     _dt(hidden_states), _stream()), "moe_route_gather")
    return topk_weights, topk_ids, sorted_ids, expert_ids, post_pad, inv, packed, m_pad


def moe_gather_pack(a: torch.Tensor, sorted_token_ids: torch.Tensor, num_tokens_post_pad: torch.Tensor,
                    m_pad: int, topk: int) -> torch.Tensor:
    """Packed (fragment-major f16) activations of the expert GEMMs: row r = a[sorted[r] // topk]."""
    lib = _lib.lib()
    k = a.shape[1]
    if a.stride(1) != 1:
        a = a.contiguous()
    out = torch.empty(lib.aphro_wna16_packed_a_bytes(m_pad, k) // 2, dtype=torch.float16, device=a.device)
    check(lib.aphro_moe_gather_pack(a.data_ptr(), sorted_token_ids.data_ptr(), num_tokens_post_pad.data_ptr(),
                                    out.data_ptr(), m_pad, k, a.stride(0), a.shape[0] * topk, topk, _dt(a),
                                    _stream()), "moe_gather_pack")
    return out


def wna16_grouped_ksplit(m_pad: int, n: int, k: int, groups: int) -> int:
    return _lib.lib().aphro_wna16_grouped_ksplit(m_pad, n, k, groups)


def wna16_gemm_grouped(a_packed: torch.Tensor, m_pad: int, k: int, qweight: torch.Tensor,
                       qzeros: torch.Tensor, scales: torch.Tensor, expert_ids: torch.Tensor,
                       num_tokens_post_pad: torch.Tensor, zero_offset: int, mode: str):
    """Grouped expert GEMM (marlin_gemm_moe role).  qweight [E, K/8, N], qzeros [E, G, N/8],
    scales [E, G, N].  mode "silu_pack" -> packed activations [m_pad, N/2]; "slabs" -> (fp32
    [S, m_pad, N], S); "out" -> [m_pad, N] in scales.dtype."""
    lib = _lib.lib()
    n = qweight.shape[2]
    groups = scales.shape[1]
    ks = lib.aphro_wna16_grouped_ksplit(m_pad, n, k, groups)
    if ks <= 0:
        raise RuntimeError(f"wna16_gemm_grouped: shape N={n} K={k} not served by the fast kernel")
    dt = _dt(scales)
    dev = qweight.device
    common = (a_packed.data_ptr(), qweight.data_ptr(), qzeros.data_ptr(), scales.data_ptr(),
              expert_ids.data_ptr(), num_tokens_post_pad.data_ptr())
    if mode == "silu_pack":
        if ks != 1 or n % 256 != 0:
            raise RuntimeError("wna16_gemm_grouped: SiluAndMul epilogue unavailable for this shape")
        out = torch.empty(lib.aphro_wna16_packed_a_bytes(m_pad, n // 2) // 2, dtype=torch.float16, device=dev)
        check(lib.aphro_wna16_gemm_grouped(*common, None, None, 0, out.data_ptr(), m_pad, n, k, groups,
                                           zero_offset, dt, _stream()), "wna16_gemm_grouped")
        return out
    if mode == "slabs":
        slabs = torch.empty((ks, m_pad, n), dtype=torch.float32, device=dev)
        check(lib.aphro_wna16_gemm_grouped(*common, None, slabs.data_ptr(), slabs.numel() * 4, None, m_pad, n, k,
                                           groups, zero_offset, dt, _stream()), "wna16_gemm_grouped")
        return slabs, ks
    out = torch.zeros((m_pad, n), dtype=scales.dtype, device=dev)
    slabs = torch.empty((ks, m_pad, n), dtype=torch.float32, device=dev) if ks > 1 else None
    check(lib.aphro_wna16_gemm_grouped(*common, out.data_ptr(), _ptr(slabs),
                                       slabs.numel() * 4 if slabs is not None else 0, None, m_pad, n, k, groups,
                                       zero_offset, dt, _stream()), "wna16_gemm_grouped")
    return out


def moe_combine(slabs: torch.Tensor, inv_pos: torch.Tensor, topk_weights: torch.Tensor,
                out_dtype: torch.dtype) -> torch.Tensor:
    """out[t] = sum_k round(w[t,k] * y[inv_pos[t,k]]), y = sum of the split-K slabs."""
    nslab, m_pad, n = slabs.shape
    t_, topk = topk_weights.shape
    out = torch.empty((t_, n), dtype=out_dtype, device=slabs.device)
    odt = _lib.F16 if out_dtype == torch.float16 else _lib.BF16
    check(_lib.lib().aphro_moe_combine(out.data_ptr(), slabs.data_ptr(), nslab, m_pad, inv_pos.data_ptr(),
                                       topk_weights.data_ptr(), t_, topk, n, odt, _stream()), "moe_combine")
    return out


def wna16_gemm(a, qweight_kpacked, qzeros, scales, perm=None, zero_offset=0):
    """The gptq_marlin_gemm role: fast W4A16 kernel on prepacked weights."""
    _require_cuda(a, qweight_kpacked, qzeros, scales)
    return _wna16(a, qweight_kpacked, qzeros, scales, perm, zero_offset)


# --------------------------------------------------------------------------
# FP8
# --------------------------------------------------------------------------
def check_fp8_buffer(t: torch.Tensor, what: str) -> None:
    """gfx950 computes in OCP e4m3fn.  The reference's ROCm branch allocates MI300's e4m3fnuz (``_custom_ops.py:663-665``,
    ``w8a8_utils.py:207-228``): the same bytes mean HALF the value there and 0x80 is NaN, so writing OCP bits into an
    fnuz-typed tensor (or multiplying fnuz-typed operands) would be silently wrong by 2x.  Refuse it."""
    fnuz = getattr(torch, "float8_e4m3fnuz", None)
    if fnuz is not None and t.dtype == fnuz:
        raise RuntimeError(f"{what}: tensor is float8_e4m3fnuz (MI300 encoding); the MI355X kernels produce / consume "
                           "OCP float8_e4m3fn -- allocate torch.float8_e4m3fn (plugin.register() patches the reference's "
                           "ROCm fp8 dtype selection)")
    if t.dtype not in (FP8_DTYPE, torch.uint8):
        raise RuntimeError(f"{what}: expected a float8_e4m3fn (or uint8) buffer, got {t.dtype}")


def scaled_fp8_quant(input: torch.Tensor, scale: Optional[torch.Tensor] = None,
                     num_token_padding: Optional[int] = None,
                     scale_ub: Optional[torch.Tensor] = None,
                     use_per_token_if_dynamic: bool = False,
                     out: Optional[torch.Tensor] = None,
                     scale_out: Optional[torch.Tensor] = None
                     ) -> Tuple[torch.Tensor, torch.Tensor]:
    """_custom_ops.py:632-685 with the OCP e4m3fn output dtype of gfx950.  ``out`` / ``scale_out``: caller-owned
    buffers (the ``Tensor!`` arguments of the schema-level ops, torch_bindings.cpp:374-390) written in place."""
    assert input.ndim == 2
    _require_cuda(input)
    if not input.is_contiguous():
        input = input.contiguous()
    lib = _lib.lib()
    shape = input.shape
    if num_token_padding:
        shape = (max(num_token_padding, input.shape[0]), shape[1])
    if out is not None:
        check_fp8_buffer(out, "scaled_fp8_quant")
        if not out.is_contiguous() or out.shape[1] != shape[1] or out.shape[0] < input.shape[0]:
            raise RuntimeError("scaled_fp8_quant: out must be a contiguous [>= M, K] buffer")
        output = out
    else:
        output = torch.empty(shape, device=input.device, dtype=FP8_DTYPE)
    m, k = input.shape
    if scale is None:
        if use_per_token_if_dynamic:
            scale = scale_out if scale_out is not None else \
                torch.empty((shape[0], 1), device=input.device, dtype=torch.float32)
            check(lib.aphro_dynamic_per_token_scaled_fp8_quant(
                output.data_ptr(), input.data_ptr(), scale.data_ptr(),
                _ptr(scale_ub), m, k, _dt(input), _stream()), "scaled_fp8_quant")
        else:
            scale = scale_out if scale_out is not None else torch.empty(1, device=input.device, dtype=torch.float32)
            ws = _quant_scratch(input.device)
            check(lib.aphro_dynamic_scaled_fp8_quant_ws(
                output.data_ptr(), input.data_ptr(), scale.data_ptr(), ws.data_ptr(), ws.numel() * 4,
                m, k, _dt(input), _stream()), "scaled_fp8_quant")
    else:
        assert scale.numel() == 1 or num_token_padding is None
        check(lib.aphro_static_scaled_fp8_quant(
            output.data_ptr(), input.data_ptr(), scale.data_ptr(), m, k,
            _dt(input), _stream()), "scaled_fp8_quant")
    return output, scale


def cutlass_scaled_mm_supports_fp8(cuda_device_capability: int) -> bool:
    return True  # gfx950 has native OCP fp8 MFMA


def cutlass_scaled_mm(a: torch.Tensor, b: torch.Tensor, scale_a: torch.Tensor,
                      scale_b: torch.Tensor, out_dtype: torch.dtype,
                      bias: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """_custom_ops.py:497-517.  a [M,K] e4m3 row-major, b [K,N] e4m3
    column-major (``weight.t()``).  ``out``: caller-owned [M,N] result buffer (the ``Tensor! out`` of the
    schema-level op, torch_bindings.cpp:235-239)."""
    _require_cuda(a, b, scale_a, scale_b)
    for x_ in (a, b):
        fnuz = getattr(torch, "float8_e4m3fnuz", None)
        if fnuz is not None and x_.dtype == fnuz:
            check_fp8_buffer(x_, "cutlass_scaled_mm")
    assert b.shape[0] % 16 == 0 and b.shape[1] % 16 == 0
    assert out_dtype in (torch.bfloat16, torch.float16)
    assert bias is None or (bias.shape[0] == b.shape[1] and bias.dtype == out_dtype)
    if a.dtype != FP8_DTYPE or b.dtype != FP8_DTYPE:
        raise RuntimeError("cutlass_scaled_mm on MI355X implements fp8 (e4m3fn) "
                           "only; int8 is out of scope (SURVEY 2.2)")
    m, k = a.shape
    n = b.shape[1]
    if b.stride(0) != 1 or b.stride(1) != k:
        raise RuntimeError("b must be column-major [K,N] (weight.t())")
    if not a.is_contiguous():
        a = a.contiguous()
    if m > SCALED_MM_LIBRARY_MIN_M and n % 128 == 0 and k % 128 == 0 and not switch("APHRO_FP8_NO_LARGE"):
        # prefill-sized M: the MFMA-bound kernel of fp8_gemm_large.hip (scales + bias in its epilogue, f16 or bf16)
        lib = _lib.lib()
        if out is None:
            out = torch.empty((m, n), dtype=out_dtype, device=a.device)
        elif out.shape != (m, n) or out.dtype != out_dtype or not out.is_contiguous():
            raise RuntimeError("cutlass_scaled_mm: out must be a contiguous [M, N] tensor of out_dtype")
        sa = scale_a.reshape(-1).float()
        sb = scale_b.reshape(-1).float()
        if sa.numel() not in (1, m) or sb.numel() not in (1, n):
            raise RuntimeError("cutlass_scaled_mm: scale_a must have 1 or M elements, scale_b 1 or N")
        ws = _workspace(a.device, lib.aphro_scaled_mm_fp8_large_workspace_bytes(m, n, k))
        check(lib.aphro_scaled_mm_fp8_large(
            out.data_ptr(), a.data_ptr(), b.data_ptr(), sa.data_ptr(), sb.data_ptr(), _ptr(bias),
            ws.data_ptr(), ws.numel(), m, n, k, 1 if sa.numel() > 1 else 0, 1 if sb.numel() > 1 else 0,
            _lib.F16 if out_dtype == torch.float16 else _lib.BF16, _stream()), "cutlass_scaled_mm")
        return out
    if m > SCALED_MM_LIBRARY_MIN_M:
        # shapes the hand-written kernel does not tile (N or K not a multiple of 128): library GEMM: a plain library GEMM (hipBLASLt through torch._scaled_mm --
        # what the reference itself calls on ROCm, w8a8_utils.py:130,165; measured 1.9 PFLOP/s fp8 at
        # M = 8192).  Row-wise scaling needs both scale vectors; a scalar one is broadcast.
        why = "APHRO_FP8_NO_LARGE is set" if (n % 128 == 0 and k % 128 == 0) else "N or K not a multiple of 128"
        _library_fallback("cutlass_scaled_mm", f"M={m} N={n} K={k}: {why} (torch._scaled_mm)")
        sa_, sb_ = scale_a.reshape(-1).float(), scale_b.reshape(-1).float()
        if sa_.numel() > 1 or sb_.numel() > 1:
            sa_ = (sa_ if sa_.numel() > 1 else sa_.expand(m)).reshape(m, 1).contiguous()
            sb_ = (sb_ if sb_.numel() > 1 else sb_.expand(n)).reshape(1, n).contiguous()
            # hipBLASLt's row-wise epilogue writes bf16 only: fp16 results are produced the way the
            # reference's own ROCm fallback does (fp32 GEMM with unit scales, then the two
            # broadcast multiplies, w8a8_utils.py:143-183)
            if out_dtype != torch.bfloat16:
                one = torch.ones((), dtype=torch.float32, device=a.device)
                acc = torch._scaled_mm(a, b, scale_a=one, scale_b=one, out_dtype=torch.float32)
                acc = acc[0] if isinstance(acc, tuple) else acc
                res = (acc * sb_ * sa_)
                if bias is not None:
                    res = res + bias
                if out is not None:
                    out.copy_(res)
                    return out
                return res.to(out_dtype)
        else:
            sa_, sb_ = sa_.reshape(()), sb_.reshape(())
        res = torch._scaled_mm(a, b, scale_a=sa_, scale_b=sb_, bias=bias, out_dtype=out_dtype)
        res = res[0] if isinstance(res, tuple) else res
        if out is not None:
            out.copy_(res)
            return out
        return res
    lib = _lib.lib()
    if out is None:
        out = torch.empty((m, n), dtype=out_dtype, device=a.device)
    elif out.shape != (m, n) or out.dtype != out_dtype or not out.is_contiguous():
        raise RuntimeError("cutlass_scaled_mm: out must be a contiguous [M, N] tensor of out_dtype")
    odt = _lib.F16 if out_dtype == torch.float16 else _lib.BF16
    sa = scale_a.reshape(-1).float()
    sb = scale_b.reshape(-1).float()
    a_tok = 1 if sa.numel() > 1 else 0
    b_ch = 1 if sb.numel() > 1 else 0
    ws = _workspace(a.device, lib.aphro_fp8_gemm_workspace_bytes(min(m, 64), n, k))
    for m0 in range(0, m, 64):
        rows = min(64, m - m0)
        sa_blk = sa[m0:] if a_tok else sa
        check(lib.aphro_scaled_mm_fp8(
            out[m0:].data_ptr(), a[m0:].data_ptr(), b.data_ptr(),
            sa_blk.data_ptr(), sb.data_ptr(), _ptr(bias), ws.data_ptr(),
            ws.numel(), rows, n, k, a_tok, b_ch, odt, _stream()),
            "cutlass_scaled_mm")
    return out


def fp8_marlin_gemm(a: torch.Tensor, b_q_weight: torch.Tensor,
                    b_scales: torch.Tensor, workspace: Optional[torch.Tensor],
                    num_bits: int, size_m: int, size_n: int, size_k: int,
                    bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """W8A16 role of _C::fp8_marlin_gemm (torch_bindings.cpp:218-222).
    b_q_weight: e4m3 [N,K] row-major (checkpoint layout, no Marlin prepack);
    b_scales: fp32 [1] or [N].  ``workspace`` (Marlin locks) is ignored."""
    _require_cuda(a, b_q_weight, b_scales)
    assert num_bits == 8
    lib = _lib.lib()
    if a.stride(1) != 1:
        a = a.contiguous()
    if size_m > WNA16_LARGE_MIN_M and size_n % 128 == 0 and size_k % 64 == 0 and not switch("APHRO_WNA16_NO_LARGE"):
        # prefill-sized M: the int4 kernel's tile machine with e4m3 weights widened in registers (wna16_gemm_large.hip)
        a_ = a[:size_m]
        if a_.stride(0) % 8 != 0 or a_.data_ptr() % 16 != 0:
            a_ = a_.contiguous()
        out = torch.empty((size_m, size_n), dtype=a.dtype, device=a.device)
        sb = b_scales.reshape(-1).float()
        nbytes = lib.aphro_fp8_w8a16_gemm_large_workspace_bytes(size_m, size_n, size_k, _dt(a))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=a.device) if nbytes else None
        check(lib.aphro_fp8_w8a16_gemm_large(
            out.data_ptr(), a_.data_ptr(), b_q_weight.data_ptr(), sb.data_ptr(), _ptr(bias), _ptr(ws), nbytes,
            size_m, size_n, size_k, a_.stride(0), 1 if sb.numel() > 1 else 0, _dt(a), _stream()), "fp8_marlin_gemm")
        return out
    if size_m >= GPTQ_DEQUANT_MIN_M:
        # shapes the hand-written kernel does not tile: widen the weight once (exact) and run a library GEMM
        _library_fallback("fp8_marlin_gemm", f"M={size_m} N={size_n} K={size_k}: shape not tiled by the W8A16 kernel (widen + matmul)")
        w = b_q_weight.to(a.dtype)
        sb_ = b_scales.reshape(-1).to(a.dtype)
        w = w * (sb_.reshape(-1, 1) if sb_.numel() > 1 else sb_)
        out = torch.matmul(a[:size_m], w.t())
        if bias is not None:
            out = out + bias
        return out
    out = torch.empty((size_m, size_n), dtype=a.dtype, device=a.device)
    sb = b_scales.reshape(-1).float()
    ws = _workspace(a.device,
                    lib.aphro_fp8_gemm_workspace_bytes(min(size_m, 64), size_n, size_k))
    for m0 in range(0, size_m, 64):
        rows = min(64, size_m - m0)
        check(lib.aphro_fp8_w8a16_gemm(
            out[m0:].data_ptr(), a[m0:].data_ptr(), b_q_weight.data_ptr(),
            sb.data_ptr(), _ptr(bias), ws.data_ptr(), ws.numel(), rows, size_n,
            size_k, a.stride(0), 1 if sb.numel() > 1 else 0, _dt(a), _stream()),
            "fp8_marlin_gemm")
    return out


# --------------------------------------------------------------------------
# glue
# --------------------------------------------------------------------------
def rms_norm(out: torch.Tensor, input: torch.Tensor, weight: torch.Tensor,
             epsilon: float) -> None:
    _require_cuda(out, input, weight)
    hidden = input.shape[-1]
    x = input.reshape(-1, hidden)
    check(_lib.lib().aphro_rms_norm(out.data_ptr(), x.data_ptr(), weight.data_ptr(),
                                    float(epsilon), x.shape[0], hidden,
                                    x.stride(0), _dt(input), _stream()), "rms_norm")


def fused_add_rms_norm(input: torch.Tensor, residual: torch.Tensor,
                       weight: torch.Tensor, epsilon: float) -> None:
    _require_cuda(input, residual, weight)
    hidden = input.shape[-1]
    assert input.is_contiguous() and residual.is_contiguous()
    check(_lib.lib().aphro_fused_add_rms_norm(
        input.data_ptr(), residual.data_ptr(), weight.data_ptr(), float(epsilon),
        input.numel() // hidden, hidden, _dt(input), _stream()),
        "fused_add_rms_norm")


def silu_and_mul(out: torch.Tensor, x: torch.Tensor, interleaved: bool = False) -> None:
    """``interleaved``: the columns of x are (gate_j, up_j) pairs (a gate_up GEMM on ops.interleave_gate_up weights)
    instead of [gate | up] halves -- same result."""
    _require_cuda(out, x)
    d = x.shape[-1] // 2
    assert x.is_contiguous() and out.is_contiguous()
    lib = _lib.lib()
    fn = lib.aphro_silu_and_mul_interleaved if interleaved else lib.aphro_silu_and_mul
    check(fn(out.data_ptr(), x.data_ptr(), x.numel() // (2 * d), d, _dt(x), _stream()), "silu_and_mul")


def rotary_embedding(positions: torch.Tensor, query: torch.Tensor,
                     key: torch.Tensor, head_size: int,
                     cos_sin_cache: torch.Tensor, is_neox: bool) -> None:
    _require_cuda(positions, query, key, cos_sin_cache)
    num_tokens = positions.numel()
    q2 = query.view(num_tokens, -1) if query.dim() != 2 else query
    k2 = key.view(num_tokens, -1) if key.dim() != 2 else key
    if positions.dtype != torch.int64:
        positions = positions.long()
    check(_lib.lib().aphro_rotary_embedding(
        positions.data_ptr(), q2.data_ptr(), k2.data_ptr(), num_tokens,
        q2.shape[1] // head_size, k2.shape[1] // head_size, head_size,
        cos_sin_cache.shape[1], cos_sin_cache.data_ptr(), q2.stride(0),
        k2.stride(0), 1 if is_neox else 0, _dt(query), _stream()),
        "rotary_embedding")


# ---------------------------------------------------------------------------
# _C_custom_ar::* (kernels/torch_bindings.cpp:506-536; aphrodite/_custom_ops.py:906-941) over
# csrc/custom_all_reduce.hip.  Same names, argument order and meaning as the reference wrappers, so that
# aphrodite/distributed/device_communicators/custom_all_reduce.py runs unchanged on top of them.
# IPC handles are the 64 opaque bytes of hipIpcMemHandle_t -- what ``storage._share_cuda_()[1]`` holds on
# ROCm -- passed as ``bytes`` (the reference passes ``bytes`` where its schema says ``str[]``); a ``str`` is
# taken as latin-1 or, when it is 128 hex digits, as hex (the only forms that survive the dispatcher's
# UTF-8 round trip: see torch_ops.CUSTOM_AR_SCHEMAS).
# ---------------------------------------------------------------------------
_CUSTOM_AR_DT = {torch.float16: _lib.F16, torch.bfloat16: _lib.BF16, torch.float32: 2}
_custom_ar_live = {}       # fa -> objects the communicator borrows (meta / rank_data tensors)


def _ipc_handle_bytes(h) -> bytes:
    n = _lib.lib().aphro_ipc_handle_bytes()
    if isinstance(h, (bytes, bytearray, memoryview)):
        b = bytes(h)
    elif isinstance(h, str):
        b = bytes.fromhex(h) if len(h) == 2 * n else h.encode("latin-1")
    else:
        raise TypeError(f"IPC handle must be bytes or str, not {type(h).__name__}")
    if len(b) != n:
        raise ValueError(f"IPC handle has {len(b)} bytes, expected {n}")
    return b


def _ipc_pack(handles, offsets):
    import ctypes
    raw = b"".join(_ipc_handle_bytes(h) for h in handles)
    return ctypes.create_string_buffer(raw, len(raw)), (ctypes.c_int64 * len(offsets))(*[int(o) for o in offsets])


def meta_size() -> int:
    """Bytes of one rank's signal area (the ``Signal`` struct, custom_all_reduce.cuh:30-42)."""
    return int(_lib.lib().aphro_custom_ar_meta_size())


class _SharedDeviceBuffer:
    """Peer-visible UNCACHED device memory exposed through __cuda_array_interface__ (freed with the object)."""

    def __init__(self, nbytes: int):
        import ctypes
        self.ptr = ctypes.c_void_p()
        check(_lib.lib().aphro_custom_ar_alloc_shared(ctypes.byref(self.ptr), nbytes), "custom_ar_alloc_shared")
        self.nbytes = nbytes
        self.__cuda_array_interface__ = {"shape": (nbytes, ), "typestr": "|u1", "data": (self.ptr.value, False),
                                         "version": 2, "strides": None}

    def __del__(self):
        try:
            if self.ptr:
                _lib.lib().aphro_custom_ar_free_shared(self.ptr)
                self.ptr = None
        except Exception:
            pass


def custom_ar_alloc_meta(nbytes: int, device) -> torch.Tensor:
    """The ``meta`` buffer of init_custom_ar (``torch.zeros(meta_size() + max_size, uint8)`` in the reference,
    custom_all_reduce.py:101-104) as zero-filled fine-grained memory: the signal words are polled by the peer GPUs
    while kernels run, which ordinary (coarse-grained, L2-cached) allocations do not guarantee to be visible."""
    with torch.cuda.device(device):
        owner = _SharedDeviceBuffer(nbytes)
        t = torch.as_tensor(owner, device=torch.device(device))
    t._aphro_owner = owner
    return t


def ipc_handle_of(t: torch.Tensor) -> Tuple[bytes, int]:
    """(handle, offset) of the allocation holding ``t`` -- ``t.untyped_storage()._share_cuda_()`` (data[1], data[3])
    in the reference (custom_all_reduce.py:210-214), taken with hipIpcGetMemHandle so that it also works for the
    fine-grained buffers above."""
    import ctypes
    lib = _lib.lib()
    h = ctypes.create_string_buffer(lib.aphro_ipc_handle_bytes())
    off = ctypes.c_int64()
    check(lib.aphro_ipc_get_mem_handle(ctypes.c_void_p(t.data_ptr()), h, ctypes.byref(off)), "ipc_get_mem_handle")
    return bytes(h.raw), int(off.value)


def init_custom_ar(meta: torch.Tensor, rank_data: torch.Tensor, handles: List, offsets: List[int], rank: int,
                   full_nvlink: bool) -> int:
    """custom_all_reduce.cu:13-41.  ``meta``: this rank's signal area followed by the two-shot scratch
    (meta_size() + max_size bytes); ``handles`` / ``offsets``: every rank's meta, in rank order; ``rank_data``: device
    memory for the registered-buffer table.  ``full_nvlink`` is accepted for signature parity: every GPU pair of an
    MI355X node is one xGMI hop, the one-shot / two-shot choice depends on the size only."""
    import ctypes
    _require_cuda(meta, rank_data)
    world = len(handles)
    if len(offsets) != world:
        raise ValueError("handles length should equal to offsets length")
    if not 0 <= rank < world:
        raise ValueError("invalid rank passed in")
    ms = meta_size()
    if meta.numel() * meta.element_size() <= ms:
        raise ValueError("meta must hold the signal area and the two-shot scratch (meta_size() + max_size bytes)")
    scratch_bytes = (meta.numel() * meta.element_size() - ms) // 16 * 16
    hbuf, obuf = _ipc_pack(handles, offsets)
    sbuf = (ctypes.c_int64 * world)(*[int(o) + ms for o in offsets])
    fa = ctypes.c_void_p()
    check(_lib.lib().aphro_custom_ar_init(
        ctypes.byref(fa), meta.data_ptr(), hbuf, obuf, meta.data_ptr() + ms, scratch_bytes, hbuf, sbuf,
        rank_data.data_ptr(), rank_data.numel() * rank_data.element_size(), rank, world), "init_custom_ar")
    _custom_ar_live[fa.value] = (meta, rank_data)
    return fa.value


def init_custom_ar_loopback(meta: torch.Tensor, rank_data: torch.Tensor, world: int) -> int:
    """A LOOPBACK communicator (aphro_custom_ar_init_loopback): ``world`` ranks that all resolve to this process's
    buffers -- the timing rig of ``bench.py --sim-tp`` (one rank of a TP group on a one-GPU box runs the real
    all-reduce kernels on local memory).  ``meta`` as in init_custom_ar."""
    import ctypes
    _require_cuda(meta, rank_data)
    ms = meta_size()
    if meta.numel() * meta.element_size() <= ms:
        raise ValueError("meta must hold the signal area and the two-shot scratch (meta_size() + max_size bytes)")
    scratch_bytes = (meta.numel() * meta.element_size() - ms) // 16 * 16
    fa = ctypes.c_void_p()
    check(_lib.lib().aphro_custom_ar_init_loopback(
        ctypes.byref(fa), meta.data_ptr(), meta.data_ptr() + ms, scratch_bytes, rank_data.data_ptr(),
        rank_data.numel() * rank_data.element_size(), world), "init_custom_ar_loopback")
    _custom_ar_live[fa.value] = (meta, rank_data)
    return fa.value


def should_one_shot(world: int, nbytes: int) -> bool:
    """The peer-access all-reduce's one-shot (every rank sums all inputs) vs two-shot choice at this message size."""
    return _lib.lib().aphro_custom_ar_should_one_shot(world, nbytes) != 0


def custom_ar_fused_norm_one_shot(world: int, tokens: int, hidden: int, esz: int = 2) -> bool:
    """True when custom_ar_fused_add_rms_norm runs its one-shot form at this size; False = the two-shot (column-slice) form."""
    return _lib.lib().aphro_custom_ar_fused_norm_one_shot(world, tokens, hidden, esz) != 0


def custom_ar_fused_add_rms_norm(fa: int, inp: torch.Tensor, residual: Optional[torch.Tensor], has_residual: bool,
                                 weight: torch.Tensor, epsilon: float, pack: bool = True, want_out: bool = False,
                                 reg_buffer: Optional[torch.Tensor] = None, prefetch: Optional[torch.Tensor] = None):
    """tensor_model_parallel_all_reduce(inp) -> fused_add_rms_norm(residual) [-> pack] in ONE launch
    (modeling/layers/linear.py:1142-1143 followed by models/llama.py's layernorm call): the bits of all_reduce_reg /
    all_reduce_unreg followed by fused_add_rms_norm_pack.  Returns (packed or None, out or None); ``residual`` is
    updated in place.  ``inp`` [tokens <= 64, hidden] must be registered (or ``reg_buffer`` given, or the stream capturing).
    ``prefetch``: the packed weights of the GEMM that follows -- extra workgroups of the launch stream them through the
    Infinity Cache while the sum waits on flags and links."""
    _require_cuda(inp, weight)
    if inp.dim() != 2 or not inp.is_contiguous() or inp.dtype != weight.dtype or inp.dtype not in (torch.float16, torch.bfloat16):
        raise RuntimeError("custom_ar_fused_add_rms_norm: inp must be a contiguous [tokens, hidden] f16 / bf16 tensor of the weight's dtype")
    tokens, hidden = inp.shape
    lib = _lib.lib()
    packed = torch.empty(lib.aphro_wna16_packed_a_bytes(tokens, hidden) // 2, dtype=torch.float16,
                         device=inp.device) if pack else None
    out = torch.empty((tokens, hidden), dtype=weight.dtype, device=inp.device) if want_out else None
    check(lib.aphro_custom_ar_fused_add_rms_norm(
        fa, inp.data_ptr(), _ptr(residual), 1 if has_residual else 0, weight.data_ptr(), float(epsilon), _ptr(packed),
        _ptr(out), tokens, hidden, _dt(weight), _ptr(prefetch),
        prefetch.numel() * prefetch.element_size() if prefetch is not None else 0, _ptr(reg_buffer),
        reg_buffer.numel() * reg_buffer.element_size() if reg_buffer is not None else 0, _stream()),
        "custom_ar_fused_add_rms_norm")
    return packed, out


def custom_ar_fused_add_rms_norm_quant_fp8(fa: int, inp: torch.Tensor, residual: Optional[torch.Tensor], has_residual: bool,
                                           weight: torch.Tensor, epsilon: float, want_out: bool = False,
                                           static_scale: Optional[torch.Tensor] = None,
                                           reg_buffer: Optional[torch.Tensor] = None):
    """tensor_model_parallel_all_reduce(inp) -> fused_add_rms_norm(residual) -> scaled_fp8_quant in ONE launch (an FP8
    W8A8 layer under TP: linear.py:1142-1143, models/llama.py's layernorm, quantization/fp8.py's activation quantisation):
    the bits of all_reduce_reg / all_reduce_unreg followed by fused_add_rms_norm_quant_fp8(inp, ...).  Returns
    (q e4m3 [tokens, hidden], scales fp32 [tokens, 1], out or None); ``residual`` is updated in place."""
    _require_cuda(inp, weight)
    if inp.dim() != 2 or not inp.is_contiguous() or inp.dtype != weight.dtype or inp.dtype not in (torch.float16, torch.bfloat16):
        raise RuntimeError("custom_ar_fused_add_rms_norm_quant_fp8: inp must be a contiguous [tokens, hidden] f16 / bf16 tensor of the weight's dtype")
    tokens, hidden = inp.shape
    q = torch.empty((tokens, hidden), dtype=FP8_DTYPE, device=inp.device)
    sc = torch.empty((tokens, 1), dtype=torch.float32, device=inp.device)
    out = torch.empty((tokens, hidden), dtype=weight.dtype, device=inp.device) if want_out else None
    _check_static_scale(static_scale, inp.device)
    check(_lib.lib().aphro_custom_ar_fused_add_rms_norm_quant_fp8(
        fa, inp.data_ptr(), _ptr(residual), 1 if has_residual else 0, weight.data_ptr(), float(epsilon), q.data_ptr(),
        sc.data_ptr(), _ptr(static_scale), _ptr(out), tokens, hidden, _dt(weight), _ptr(reg_buffer),
        reg_buffer.numel() * reg_buffer.element_size() if reg_buffer is not None else 0, _stream()),
        "custom_ar_fused_add_rms_norm_quant_fp8")
    return q, sc, out


def custom_ar_fused_add_rms_norm_router(fa: int, inp: torch.Tensor, residual: Optional[torch.Tensor], has_residual: bool,
                                        weight: torch.Tensor, epsilon: float, router_weight: torch.Tensor,
                                        reg_buffer: Optional[torch.Tensor] = None):
    """tensor_model_parallel_all_reduce(inp) -> fused_add_rms_norm(residual) -> the router's logits of a sparse-MLP layer in
    ONE launch (linear.py:1142-1143, models/mixtral.py's post_attention_layernorm and MixtralMoE.gate, mixtral.py:60-110):
    the bits of all_reduce_reg / all_reduce_unreg followed by fused_add_rms_norm_router(inp, ...).  Returns
    (out [tokens, hidden], router_logits [tokens, E]), E <= 16; ``residual`` is updated in place."""
    _require_cuda(inp, weight, router_weight)
    if inp.dim() != 2 or not inp.is_contiguous() or inp.dtype != weight.dtype or inp.dtype not in (torch.float16, torch.bfloat16):
        raise RuntimeError("custom_ar_fused_add_rms_norm_router: inp must be a contiguous [tokens, hidden] f16 / bf16 tensor of the weight's dtype")
    tokens, hidden = inp.shape
    e = router_weight.shape[0]
    if router_weight.shape[1] != hidden or not router_weight.is_contiguous() or router_weight.dtype != weight.dtype:
        raise RuntimeError("custom_ar_fused_add_rms_norm_router: router_weight must be a contiguous [E, hidden] tensor of the norm's dtype")
    out = torch.empty((tokens, hidden), dtype=weight.dtype, device=inp.device)
    logits = torch.empty((tokens, e), dtype=weight.dtype, device=inp.device)
    check(_lib.lib().aphro_custom_ar_fused_add_rms_norm_router(
        fa, inp.data_ptr(), _ptr(residual), 1 if has_residual else 0, weight.data_ptr(), float(epsilon), out.data_ptr(),
        router_weight.data_ptr(), logits.data_ptr(), e, tokens, hidden, _dt(weight), _ptr(reg_buffer),
        reg_buffer.numel() * reg_buffer.element_size() if reg_buffer is not None else 0, _stream()),
        "custom_ar_fused_add_rms_norm_router")
    return out, logits


def _ar_check_io(inp: torch.Tensor, out: torch.Tensor):
    _require_cuda(inp, out)
    if inp.dtype != out.dtype or inp.numel() != out.numel():
        raise RuntimeError("all_reduce: inp and out must have the same dtype and number of elements")
    if inp.dtype not in _CUSTOM_AR_DT:
        raise RuntimeError("custom allreduce only supports float32, float16 and bfloat16")


def all_reduce_reg(fa: int, inp: torch.Tensor, out: torch.Tensor) -> None:
    """custom_all_reduce.cu:84-92: ``inp`` is a registered buffer (or is being captured into a graph)."""
    _ar_check_io(inp, out)
    check(_lib.lib().aphro_custom_ar_all_reduce(fa, inp.data_ptr(), out.data_ptr(), inp.numel(), _CUSTOM_AR_DT[inp.dtype],
                                                None, 0, _stream()), "all_reduce_reg")


def all_reduce_unreg(fa: int, inp: torch.Tensor, reg_buffer: torch.Tensor, out: torch.Tensor) -> None:
    """custom_all_reduce.cu:94-109: copy ``inp`` into the registered ``reg_buffer``, then reduce from there."""
    _ar_check_io(inp, out)
    nbytes = inp.numel() * inp.element_size()
    if nbytes > reg_buffer.numel() * reg_buffer.element_size():
        raise RuntimeError("registered buffer is too small to contain the input")
    check(_lib.lib().aphro_custom_ar_all_reduce(fa, inp.data_ptr(), out.data_ptr(), inp.numel(), _CUSTOM_AR_DT[inp.dtype],
                                                reg_buffer.data_ptr(), reg_buffer.numel() * reg_buffer.element_size(),
                                                _stream()), "all_reduce_unreg")


def dispose(fa: int) -> None:
    _lib.lib().aphro_custom_ar_dispose(fa)
    _custom_ar_live.pop(fa, None)


def register_buffer(fa: int, t: torch.Tensor, handles: List, offsets: List[int]) -> None:
    hbuf, obuf = _ipc_pack(handles, offsets)
    check(_lib.lib().aphro_custom_ar_register_buffer(fa, t.data_ptr(), hbuf, obuf), "register_buffer")


def get_graph_buffer_ipc_meta(fa: int) -> Tuple[bytes, List[int]]:
    """(handle blob, offsets) of the buffers all-reduced while the last graph was captured: the blob is the buffers'
    IPC handles back to back -- ``std::vector<uint8_t>`` in the reference (custom_all_reduce.cu:120-128), which its
    caller turns into ``bytes(handle)`` (custom_all_reduce.py:232-233)."""
    import ctypes
    lib = _lib.lib()
    hb = lib.aphro_ipc_handle_bytes()
    n = ctypes.c_int()
    check(lib.aphro_custom_ar_get_graph_buffer_ipc_meta(fa, None, None, 0, ctypes.byref(n)), "get_graph_buffer_ipc_meta")
    count = n.value
    if count == 0:
        return b"", []
    hbuf = ctypes.create_string_buffer(count * hb)
    obuf = (ctypes.c_int64 * count)()
    check(lib.aphro_custom_ar_get_graph_buffer_ipc_meta(fa, hbuf, obuf, count, ctypes.byref(n)), "get_graph_buffer_ipc_meta")
    return bytes(hbuf.raw[:count * hb]), [int(o) for o in obuf]


def register_graph_buffers(fa: int, handles: List, offsets: List[List[int]]) -> None:
    """``handles[r]`` / ``offsets[r]``: rank r's get_graph_buffer_ipc_meta -- the handle blob (bytes; a str is taken as
    hex or latin-1) or a list of single handles (custom_all_reduce.cu:130-137)."""
    import ctypes
    hb = _lib.lib().aphro_ipc_handle_bytes()
    world = len(handles)
    if len(offsets) != world:
        raise RuntimeError("register_graph_buffers: handles and offsets must have one entry per rank")
    count = len(offsets[0]) if world else 0
    blobs = []
    for hr, orow in zip(handles, offsets):
        if isinstance(hr, (list, tuple)):
            blob = b"".join(_ipc_handle_bytes(h) for h in hr)
        elif isinstance(hr, str):
            blob = bytes.fromhex(hr) if len(hr) == 2 * hb * len(orow) else hr.encode("latin-1")
        else:
            blob = bytes(hr)
        if len(orow) != count or len(blob) != count * hb:
            raise RuntimeError("register_graph_buffers: every rank must contribute the same number of buffers")
        blobs.append(blob)
    raw = b"".join(blobs)                                                        # rank-major [world][count]
    flat = [int(o) for orow in offsets for o in orow]
    check(_lib.lib().aphro_custom_ar_register_graph_buffers(
        fa, ctypes.create_string_buffer(raw, max(1, len(raw))), (ctypes.c_int64 * max(1, len(flat)))(*flat), count),
        "register_graph_buffers")


def custom_ar_error(fa: int) -> bool:
    """True if one of this rank's barriers timed out since the last query (bounded spin instead of a hung GPU)."""
    return _lib.lib().aphro_custom_ar_error(fa) != 0
