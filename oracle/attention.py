"""Oracle: paged KV cache write, paged-attention decode, varlen causal prefill.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference anchors (relative to /root/reference):
  cache layout            aphrodite/attention/ops/paged_attn.py:40-62
  reshape_and_cache       kernels/cache_kernels.cu:152-204,
                          tests/kernels/test_cache.py:176-192
  decode attention        tests/kernels/test_attention.py:46-114 (torch ref),
                          kernels/attention/attention_kernels.cu:87-496
                          (alibi :295-298, masking :303-306, 1/(sum+1e-6)
                          :336-346, v2 partials :350-358, reduce :564-669)
  prefill (varlen causal) aphrodite/attention/backends/rocm_flash_attn.py:598-630
  prefill with context    aphrodite/attention/ops/prefix_prefill.py:58-255
"""
import numpy as np

from . import fp8 as _fp8


def _f32(x):
    """numpy float32 view of fp16/bf16/fp32 inputs (bf16 arrives as torch)."""
    try:
        import torch
        if isinstance(x, torch.Tensor):
            return x.detach().to(torch.float32).cpu().numpy()
    except ImportError:  # pragma: no cover
        pass
    return np.asarray(x).astype(np.float32)


def split_kv_cache_shapes(num_blocks, num_kv_heads, head_size, block_size,
                          elem_size):
    """paged_attn.py:49-62: K [NB,Hkv,hd/x,block,x] with x = 16/elem_size,
    V [NB,Hkv,hd,block]."""
    x = 16 // elem_size
    return ((num_blocks, num_kv_heads, head_size // x, block_size, x),
            (num_blocks, num_kv_heads, head_size, block_size))


def reshape_and_cache(key, value, key_cache, value_cache, slot_mapping,
                      kv_cache_dtype="auto", k_scale=1.0, v_scale=1.0):
    """In-place scatter of new tokens (cache_kernels.cu:152-204).
    key/value [T,Hkv,hd] float; caches numpy arrays in the layouts above
    (uint8 for fp8).  slot < 0 is padding and is skipped."""
    key = _f32(key)
    value = _f32(value)
    nb, hkv, hdx, bs, x = key_cache.shape
    for t, slot in enumerate(np.asarray(slot_mapping).tolist()):
        if slot < 0:
            continue
        b, off = divmod(int(slot), bs)
        k = key[t].reshape(hkv, hdx, x)
        v = value[t]
        if kv_cache_dtype == "auto":
            key_cache[b, :, :, off, :] = k.astype(key_cache.dtype)
            value_cache[b, :, :, off] = v.astype(value_cache.dtype)
        else:
            key_cache[b, :, :, off, :] = _fp8.kv_quant(k, k_scale, kv_cache_dtype)
            value_cache[b, :, :, off] = _fp8.kv_quant(v, v_scale, kv_cache_dtype)


def reshape_and_cache_flash(key, value, key_cache, value_cache, slot_mapping,
                            kv_cache_dtype="auto", k_scale=1.0, v_scale=1.0):
    """Flash-layout cache write: caches [NB, block, H, hd] (cache_kernels.cu:207-250,
    tests/kernels/test_cache.py:268-292)."""
    key = _f32(key)
    value = _f32(value)
    bs = key_cache.shape[1]
    for t, slot in enumerate(np.asarray(slot_mapping).tolist()):
        if slot < 0:
            continue
        b, off = divmod(int(slot), bs)
        if kv_cache_dtype == "auto":
            key_cache[b, off] = key[t].astype(key_cache.dtype)
            value_cache[b, off] = value[t].astype(value_cache.dtype)
        else:
            key_cache[b, off] = _fp8.kv_quant(key[t], k_scale, kv_cache_dtype)
            value_cache[b, off] = _fp8.kv_quant(value[t], v_scale, kv_cache_dtype)


def copy_blocks(key_caches, value_caches, block_mapping):
    """Per layer: cache[dst] = cache[src] for every (src, dst) pair, in order
    (cache_kernels.cu:66-100; tests/kernels/test_cache.py:84-90)."""
    for src, dst in np.asarray(block_mapping).reshape(-1, 2).tolist():
        for kc in key_caches:
            kc[dst] = kc[src]
        for vc in value_caches:
            vc[dst] = vc[src]


def swap_blocks(src, dst, block_mapping):
    """dst[d] = src[s] for every (s, d) pair (cache_kernels.cu:24-63;
    tests/kernels/test_cache.py:383-388)."""
    for s_, d_ in np.asarray(block_mapping).reshape(-1, 2).tolist():
        dst[d_] = src[s_]


def advance_step(input_tokens, sampled_token_ids, input_positions, seq_lens, slot_mapping,
                 block_tables, block_size, num_queries):
    """prepare_inputs/advance_step.cu:13-51: in place, for the first num_queries rows."""
    for q in range(num_queries):
        input_tokens[q] = sampled_token_ids[q]
        nxt = int(seq_lens[q]) + 1
        pos = nxt - 1
        seq_lens[q] = nxt
        input_positions[q] = pos
        slot_mapping[q] = int(block_tables[q][pos // block_size]) * block_size + pos % block_size


def gather_kv(key_cache, value_cache, block_table, seq_len,
              kv_cache_dtype="auto", k_scale=1.0, v_scale=1.0):
    """-> K,V float32 [L,Hkv,hd] for one sequence (test_attention.py:81-93)."""
    nb, hkv, hdx, bs, x = key_cache.shape
    pos = np.arange(seq_len)
    blk = np.asarray(block_table)[pos // bs].astype(np.int64)
    off = pos % bs
    k = key_cache[blk, :, :, off, :]           # [L,Hkv,hd/x,x]
    v = value_cache[blk, :, :, off]            # [L,Hkv,hd]
    k = k.reshape(seq_len, hkv, hdx * x)
    if kv_cache_dtype == "auto":
        return _f32(k), _f32(v)
    return (_fp8.kv_dequant(k, k_scale, kv_cache_dtype),
            _fp8.kv_dequant(v, v_scale, kv_cache_dtype))


def paged_attention_decode(query, key_cache, value_cache, block_tables,
                           seq_lens, scale, alibi_slopes=None,
                           kv_cache_dtype="auto", k_scale=1.0, v_scale=1.0):
    """out[s,h,:] = softmax(scale*q.K^T + alibi) V over the sequence's paged KV,
    float64 math.  query [S,Hq,hd]."""
    q = _f32(query).astype(np.float64)
    num_seqs, hq, hd = q.shape
    hkv = value_cache.shape[1]
    rep = hq // hkv
    out = np.zeros((num_seqs, hq, hd), dtype=np.float64)
    for i in range(num_seqs):
        L = int(seq_lens[i])
        if L == 0:
            continue
        k, v = gather_kv(key_cache, value_cache, block_tables[i], L,
                         kv_cache_dtype, k_scale, v_scale)
        k = np.repeat(k.astype(np.float64), rep, axis=1)   # [L,Hq,hd]
        v = np.repeat(v.astype(np.float64), rep, axis=1)
        logits = scale * np.einsum("hd,lhd->hl", q[i], k)
        if alibi_slopes is not None:
            bias = (np.arange(L) - L + 1).astype(np.float64)
            logits = logits + np.asarray(alibi_slopes, np.float64)[:, None] * bias
        logits -= logits.max(axis=1, keepdims=True)
        p = np.exp(logits)
        p /= p.sum(axis=1, keepdims=True)
        out[i] = np.einsum("hl,lhd->hd", p, v)
    return out


def paged_attention_v2_partials(query, key_cache, value_cache, block_tables,
                                seq_lens, scale, partition_size=512,
                                alibi_slopes=None, kv_cache_dtype="auto",
                                k_scale=1.0, v_scale=1.0):
    """Per-partition (max_logits, exp_sums, tmp_out) exactly as
    attention_kernels.cu:320-358 defines them, and the merged output by
    :564-669.  Partitions past a sequence's end are left as NaN."""
    q = _f32(query).astype(np.float64)
    num_seqs, hq, hd = q.shape
    hkv = value_cache.shape[1]
    rep = hq // hkv
    max_len = int(max(seq_lens)) if len(seq_lens) else 0
    parts = max(1, -(-max_len // partition_size))
    mx = np.full((num_seqs, hq, parts), np.nan)
    es = np.full((num_seqs, hq, parts), np.nan)
    tmp = np.full((num_seqs, hq, parts, hd), np.nan)
    out = np.zeros((num_seqs, hq, hd))
    for i in range(num_seqs):
        L = int(seq_lens[i])
        if L == 0:
            continue
        k, v = gather_kv(key_cache, value_cache, block_tables[i], L,
                         kv_cache_dtype, k_scale, v_scale)
        k = np.repeat(k.astype(np.float64), rep, axis=1)
        v = np.repeat(v.astype(np.float64), rep, axis=1)
        logits = scale * np.einsum("hd,lhd->hl", q[i], k)
        if alibi_slopes is not None:
            bias = (np.arange(L) - L + 1).astype(np.float64)
            logits = logits + np.asarray(alibi_slopes, np.float64)[:, None] * bias
        np_i = -(-L // partition_size)
        for p in range(np_i):
            sl = slice(p * partition_size, min(L, (p + 1) * partition_size))
            lg = logits[:, sl]
            m = lg.max(axis=1)
            e = np.exp(lg - m[:, None])
            s = e.sum(axis=1)
            mx[i, :, p] = m
            es[i, :, p] = s
            tmp[i, :, p] = np.einsum("hl,lhd->hd", e / (s[:, None] + 1e-6), v[sl])
        gm = mx[i, :, :np_i].max(axis=1)
        r = es[i, :, :np_i] * np.exp(mx[i, :, :np_i] - gm[:, None])
        inv = 1.0 / (r.sum(axis=1) + 1e-6)
        out[i] = np.einsum("hp,hpd->hd", r * inv[:, None], tmp[i, :, :np_i])
    return out, mx, es, tmp


def varlen_causal_attention(q, k, v, cu_seqlens, scale, causal=True, alibi_slopes=None, window_left=None):
    """Prefill self-attention over packed sequences (rocm_flash_attn.py:598-630).
    q [T,Hq,hd], k/v [T,Hkv,hd]; float64 math.  alibi bias = slope_h * (key_pos - query_pos)
    (_make_alibi_bias, rocm_flash_attn.py:238-263).  window_left: the `left` of flash_attn_varlen_func's window_size --
    the call at rocm_flash_attn.py:497-507 passes (sliding_window, sliding_window) with causal=True.  flash_attn is a
    third-party dependency that is not vendored in the reference (requirements-rocm: the ROCm CK fork, unpinned); its
    published mask: query i (aligned to the end of the keys) sees keys j with i - left <= j <= i + right, right = 0 under
    causal -- keys i - left .. i."""
    q = _f32(q).astype(np.float64)
    k = _f32(k).astype(np.float64)
    v = _f32(v).astype(np.float64)
    hq, hkv = q.shape[1], k.shape[1]
    rep = hq // hkv
    out = np.zeros_like(q)
    cu = np.asarray(cu_seqlens).tolist()
    for b in range(len(cu) - 1):
        s, e = cu[b], cu[b + 1]
        n = e - s
        if n == 0:
            continue
        kk = np.repeat(k[s:e], rep, axis=1)
        vv = np.repeat(v[s:e], rep, axis=1)
        lg = scale * np.einsum("qhd,khd->hqk", q[s:e], kk)
        if alibi_slopes is not None:
            rel = np.arange(n)[None, :] - np.arange(n)[:, None]        # key_pos - query_pos
            lg = lg + np.asarray(alibi_slopes, np.float64)[:, None, None] * rel[None]
        if causal:
            mask = np.triu(np.ones((n, n), dtype=bool), 1)
            lg = np.where(mask[None], -np.inf, lg)
        if window_left is not None and window_left >= 0:
            far = (np.arange(n)[:, None] - np.arange(n)[None, :]) > window_left      # query_pos - key_pos > left
            lg = np.where(far[None], -np.inf, lg)
        lg -= lg.max(axis=2, keepdims=True)
        p = np.exp(lg)
        p /= p.sum(axis=2, keepdims=True)
        out[s:e] = np.einsum("hqk,khd->qhd", p, vv)
    return out


def context_attention(q, k, v, key_cache, value_cache, block_tables,
                      query_start_loc, seq_lens, ctx_lens, scale,
                      kv_cache_dtype="auto", k_scale=1.0, v_scale=1.0,
                      alibi_slopes=None, sliding_window=0, cache_round=None):
    """Prefill with cached context (prefix_prefill.py:58-255): each query token
    attends to the sequence's cached context (paged) plus the causal part of
    the new tokens.  seq_lens = ctx_len + query_len.  alibi bias =
    slope * (key_pos - query_pos) (_fwd_kernel_alibi, :460-690); sliding window
    masks keys with query_pos - key_pos >= window (:137-150).  cache_round:
    optional callable rounding the dequantised cache values to the query dtype
    (the kernel's `.to(q.dtype)`, :131-134)."""
    q = _f32(q).astype(np.float64)
    kn = _f32(k).astype(np.float64)
    vn = _f32(v).astype(np.float64)
    hq, hkv = q.shape[1], kn.shape[1]
    rep = hq // hkv
    out = np.zeros_like(q)
    for b in range(len(seq_lens)):
        s = int(query_start_loc[b])
        c = int(ctx_lens[b])
        n = int(seq_lens[b]) - c
        if n <= 0:
            continue
        if c > 0:
            kc, vc = gather_kv(key_cache, value_cache, block_tables[b], c,
                               kv_cache_dtype, k_scale, v_scale)
            if cache_round is not None:
                kc, vc = cache_round(kc), cache_round(vc)
            kk = np.concatenate([kc.astype(np.float64), kn[s:s + n]], 0)
            vv = np.concatenate([vc.astype(np.float64), vn[s:s + n]], 0)
        else:
            kk, vv = kn[s:s + n], vn[s:s + n]
        kk = np.repeat(kk, rep, axis=1)
        vv = np.repeat(vv, rep, axis=1)
        lg = scale * np.einsum("qhd,khd->hqk", q[s:s + n], kk)
        qpos = c + np.arange(n)[:, None]
        kpos = np.arange(c + n)[None, :]
        if alibi_slopes is not None:
            lg = lg + np.asarray(alibi_slopes, np.float64)[:, None, None] * (kpos - qpos)[None]
        mask = kpos > qpos
        if sliding_window and sliding_window > 0:
            mask = mask | ((qpos - kpos) >= sliding_window)
        lg = np.where(mask[None], -np.inf, lg)
        lg -= lg.max(axis=2, keepdims=True)
        p = np.exp(lg)
        p /= p.sum(axis=2, keepdims=True)
        out[s:s + n] = np.einsum("hqk,khd->qhd", p, vv)
    return out


# -- glue ops between the hot kernels ------------
def rms_norm(x, weight, eps):
    """kernels/layernorm_kernels.cu:17-45."""
    x = _f32(x).astype(np.float64)
    var = (x * x).mean(axis=-1, keepdims=True)
    return x / np.sqrt(var + eps) * _f32(weight).astype(np.float64)


def fused_add_rms_norm(x, residual, weight, eps):
    """layernorm_kernels.cu:200-240: residual' = x + residual (rounded to the
    storage dtype by the kernel; the caller rounds), out = rms_norm(residual')."""
    r = _f32(x).astype(np.float64) + _f32(residual).astype(np.float64)
    return rms_norm(r, weight, eps), r


def silu_and_mul(x):
    """kernels/activation_kernels.cu:12-60: silu(x[..., :d]) * x[..., d:]."""
    x = _f32(x).astype(np.float64)
    d = x.shape[-1] // 2
    a, b = x[..., :d], x[..., d:]
    return a / (1.0 + np.exp(-a)) * b


def rotary_embedding_neox(positions, query, key, head_size, cos_sin_cache):
    """kernels/pos_encoding_kernels.cu:10-80 (is_neox=True, rot_dim=head_size).
    query [T, Hq*hd], key [T, Hkv*hd]; cos_sin_cache [max_pos, rot_dim]
    (cos first half, sin second half)."""
    cs = _f32(cos_sin_cache).astype(np.float64)
    rot = cs.shape[1]
    half = rot // 2
    pos = np.asarray(positions).astype(np.int64)
    cos = cs[pos, :half][:, None, :]
    sin = cs[pos, half:][:, None, :]

    def rope(t):
        t = _f32(t).astype(np.float64)
        T = t.shape[0]
        t = t.reshape(T, -1, head_size).copy()
        x1 = t[..., :half].copy()
        x2 = t[..., half:rot].copy()
        t[..., :half] = x1 * cos - x2 * sin
        t[..., half:rot] = x2 * cos + x1 * sin
        return t.reshape(T, -1)

    return rope(query), rope(key)
