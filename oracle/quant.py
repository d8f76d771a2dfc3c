"""Oracle: int4 weight formats (GPTQ / AWQ) -- pack, unpack, shuffle, dequant, GEMM.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference anchors (relative to /root/reference):
  quantize_weights      aphrodite/quantization/utils/quant_utils.py:123-211
  permute_rows/sort     quant_utils.py:96-119, 313-331
  gptq_pack (pack_rows) quant_utils.py:334-355, 414-421
  awq_pack  (pack_cols) quant_utils.py:358-380, 423-441
  unpack_cols           quant_utils.py:383-411
  exllama nibble order  kernels/quantization/gptq/qdq_4.cuh:13-35
  make_sequential       kernels/quantization/gptq/q_gemm.cu:1621-1657
  zero + 1 rule         q_gemm.cu:266 (exllama), :1427 (reconstruct_gptq)
  qzeros nibble order   kernels/quantization/gptq/matrix_view.cuh:103-125
  AWQ dequant           tests/kernels/test_awq_triton.py:13-56,
                        kernels/quantization/awq/gemm_kernels.cu:340-403
"""
import numpy as np

AWQ_ORDER = np.array([0, 2, 4, 6, 1, 3, 5, 7])          # quant_utils.py:433
AWQ_REVERSE_ORDER = np.array([0, 4, 1, 5, 2, 6, 3, 7])  # test_awq_triton.py:15


# --------------------------------------------------------------------------
# quantisation of a float weight (test-input generator, mirrors the reference)
# --------------------------------------------------------------------------
def quantize_weights(w, num_bits=4, group_size=128, zero_points=False,
                     bias=None):
    """quant_utils.py:123-211.  w: float [K,N] (float32 numpy).

    Returns (w_ref, w_q, w_s, w_zp).  Symmetric types carry a bias
    (uint4b8 -> bias 8): w_q = round(w/s) + bias.  With zero points
    (uint4): w_q = round(w/s) + zp.
    Scales are rounded to float16 like the reference (orig_type half).
    """
    w = np.asarray(w)
    dt = w.dtype                      # every step stays in the input dtype
    assert dt in (np.float16, np.float32)
    size_k, size_n = w.shape
    if group_size == -1:
        group_size = size_k
    max_q = (1 << num_bits) - 1
    if zero_points:
        min_q_val, max_q_val = 0, max_q
        bias = 0
    else:
        if bias is None:
            bias = 1 << (num_bits - 1)
        min_q_val, max_q_val = -bias, max_q - bias
    wg = w.reshape(-1, group_size, size_n)
    mx = wg.max(axis=1)
    mn = wg.min(axis=1)
    with np.errstate(divide="ignore", invalid="ignore"):
        if zero_points:
            w_s = (np.maximum(mx - mn, dt.type(1e-5)) / dt.type(max_q_val))
            w_s = w_s.astype(dt)
            zp = np.clip(np.round(np.abs(mn / w_s)), min_q_val, max_q_val)
            zp = zp.astype(np.int32)
        else:
            a = (np.abs(mx / dt.type(max_q_val)) if max_q_val != 0 else
                 np.zeros_like(mx))
            b = (np.abs(mn / dt.type(min_q_val)) if min_q_val != 0 else
                 np.zeros_like(mn))
            w_s = np.maximum(a, b).astype(dt)
            zp = None
        q = np.round((wg / w_s[:, None, :]).astype(dt)).astype(np.int32)
    if zero_points:
        q = q + zp[:, None, :]
    q = np.clip(q, min_q_val, max_q_val)
    if zero_points:
        w_ref = (q - zp[:, None, :]).astype(dt) * w_s[:, None, :]
    else:
        w_ref = q.astype(dt) * w_s[:, None, :]
    q = q + bias
    return (w_ref.astype(dt).reshape(size_k, size_n),
            q.reshape(size_k, size_n), w_s, zp)


# --------------------------------------------------------------------------
# GPTQ (AutoGPTQ v1) tensors
# --------------------------------------------------------------------------
def _pack_bitstring(q, num_bits, axis):
    """Values laid end to end, little-endian, into uint32 words along `axis` -- what AutoGPTQ does for every width; for
    3 bits 32 values fill 3 words and values 10 / 21 straddle a word boundary (the reference reads them back exactly so:
    reconstruct_gptq_3bit_kernel, q_gemm.cu:1437-1478; MatrixView_q3_row, matrix_view.cuh:185-212)."""
    q = np.moveaxis(np.asarray(q).astype(np.uint64), axis, 0)
    n = q.shape[0]
    assert (n * num_bits) % 32 == 0
    bits = ((q[:, None, ...] >> np.arange(num_bits, dtype=np.uint64).reshape((1, num_bits) + (1, ) * (q.ndim - 1))) & 1)
    bits = bits.reshape((n * num_bits // 32, 32) + q.shape[1:]).astype(np.uint64)
    words = (bits << np.arange(32, dtype=np.uint64).reshape((1, 32) + (1, ) * (q.ndim - 1))).sum(axis=1)
    return np.moveaxis(words.astype(np.uint32), 0, axis)


def _unpack_bitstring(words, num_bits, axis):
    w = np.moveaxis(np.asarray(words).view(np.uint32).astype(np.uint64), axis, 0)
    nw = w.shape[0]
    bits = (w[:, None, ...] >> np.arange(32, dtype=np.uint64).reshape((1, 32) + (1, ) * (w.ndim - 1))) & 1
    bits = bits.reshape((nw * 32 // num_bits, num_bits) + w.shape[1:])
    vals = (bits << np.arange(num_bits, dtype=np.uint64).reshape((1, num_bits) + (1, ) * (w.ndim - 1))).sum(axis=1)
    return np.moveaxis(vals.astype(np.int32), 0, axis)


def gptq_pack(q_w, num_bits=4):
    """pack_rows, quant_utils.py:334-355: nibble i of word r = element 8r+i."""
    if 32 % num_bits != 0:
        return _pack_bitstring(q_w, num_bits, 0).view(np.int32)
    q_w = np.asarray(q_w).astype(np.uint32)
    pf = 32 // num_bits
    size_k, size_n = q_w.shape
    assert size_k % pf == 0
    res = np.zeros((size_k // pf, size_n), dtype=np.uint32)
    for i in range(pf):
        res |= q_w[i::pf, :] << np.uint32(num_bits * i)
    return res.view(np.int32)


def gptq_unpack(qweight, num_bits=4):
    """Inverse of gptq_pack -> int32 [K,N]."""
    if 32 % num_bits != 0:
        return _unpack_bitstring(qweight, num_bits, 0)
    qw = np.asarray(qweight).view(np.uint32)
    pf = 32 // num_bits
    rows, size_n = qw.shape
    out = np.zeros((rows * pf, size_n), dtype=np.int32)
    mask = np.uint32((1 << num_bits) - 1)
    for i in range(pf):
        out[i::pf, :] = ((qw >> np.uint32(num_bits * i)) & mask).astype(np.int32)
    return out


def pack_cols(q_w, num_bits=4):
    if 32 % num_bits != 0:
        return _pack_bitstring(q_w, num_bits, 1).view(np.int32)
    return _pack_cols_pow2(q_w, num_bits)


def _pack_cols_pow2(q_w, num_bits=4):
    """quant_utils.py:358-380: nibble i of word c = column 8c+i."""
    q_w = np.asarray(q_w).astype(np.uint32)
    pf = 32 // num_bits
    size_k, size_n = q_w.shape
    assert size_n % pf == 0
    res = np.zeros((size_k, size_n // pf), dtype=np.uint32)
    for i in range(pf):
        res |= q_w[:, i::pf] << np.uint32(num_bits * i)
    return res.view(np.int32)


def unpack_cols(packed, num_bits=4):
    if 32 % num_bits != 0:
        return _unpack_bitstring(packed, num_bits, 1)
    return _unpack_cols_pow2(packed, num_bits)


def _unpack_cols_pow2(packed, num_bits=4):
    """quant_utils.py:383-411."""
    p = np.asarray(packed).view(np.uint32)
    pf = 32 // num_bits
    size_k, words = p.shape
    out = np.zeros((size_k, words * pf), dtype=np.int32)
    mask = np.uint32((1 << num_bits) - 1)
    for i in range(pf):
        out[:, i::pf] = ((p >> np.uint32(num_bits * i)) & mask).astype(np.int32)
    return out


def gptq_pack_zeros(zeros, num_bits=4):
    """AutoGPTQ v1 qzeros [G, N/8]: stored value = zero - 1 (matrix_view.cuh:
    103-107 reads nibble (col & 7) of word col/8; q_gemm.cu:266 adds the 1)."""
    z = (np.asarray(zeros).astype(np.int64) - 1) & ((1 << num_bits) - 1)
    return pack_cols(z, num_bits)


def gptq_unpack_zeros(qzeros, num_bits=4):
    """-> the *effective* zero point (stored + 1) as int32 [G,N]."""
    return unpack_cols(qzeros, num_bits) + 1


def shuffle_4bit_word(q):
    """qdq_4.cuh:17-35 on uint32 arrays: elements 0,2,4,6 -> bits[15:0],
    elements 1,3,5,7 -> bits[31:16] ("77775555 33331111 66664444 22220000"
    is the source-nibble labelling of that transform)."""
    qa = np.asarray(q).astype(np.uint32).copy()
    qb = np.zeros_like(qa)
    for i in range(4):
        qa0 = qa & np.uint32(0x0F)
        qa1 = (qa & np.uint32(0xF0)) >> np.uint32(4)
        qa = qa >> np.uint32(8)
        qb |= qa1 << np.uint32(i * 4 + 16)
        qb |= qa0 << np.uint32(i * 4)
    return qb


def unshuffle_4bit_word(q):
    """Inverse of shuffle_4bit_word."""
    q = np.asarray(q).astype(np.uint32)
    out = np.zeros_like(q)
    for j in range(8):
        src = (j // 2) * 4 + (16 if j % 2 else 0)
        out |= ((q >> np.uint32(src)) & np.uint32(0xF)) << np.uint32(4 * j)
    return out


def make_sequential_4bit(qweight, q_perm):
    """q_gemm.cu:1621-1657: new row r (k = 8r+i) takes source k = q_perm[8r+i]."""
    w = gptq_unpack(qweight)              # [K,N]
    q_perm = np.asarray(q_perm).astype(np.int64)
    return gptq_pack(w[q_perm, :])


def gptq_shuffle(qweight, q_perm=None, bits=4):
    """ops.gptq_shuffle (q_gemm.cu:1826-1860, 4-bit): optional row permutation
    (act-order) followed by the per-word nibble shuffle.  Returns a new int32
    array (the op mutates in place).

    bits 2 / 3 / 8: the layout AFTER the shuffle is private to the kernels that
    consume it (the reference's shuffle_{2,3,8}bit_kernel reorder the fields inside
    a word for ITS dequant routines); this package keeps the checkpoint's
    sequential bitstring and only makes act-order rows sequential
    (make_sequential_{2,3,8}bit_kernel's role, q_gemm.cu:1659-1820)."""
    if bits != 4:
        qw = np.asarray(qweight)
        if q_perm is not None and len(q_perm) > 0:
            return gptq_pack(gptq_unpack(qw, bits)[np.asarray(q_perm).astype(np.int64), :], bits)
        return qw.copy()
    qw = np.asarray(qweight)
    if q_perm is not None and len(q_perm) > 0:
        qw = make_sequential_4bit(qw, q_perm)
    return shuffle_4bit_word(qw.view(np.uint32)).view(np.int32)


def gptq_dequant(qweight, qzeros, scales, g_idx=None, shuffled=False,
                 group_size=None, bits=4):
    """W[K,N] float32 = (q - (z_stored + 1)) * s   (q_gemm.cu:1394-1434).

    qweight int32 [K/8,N] (AutoGPTQ order, or exllama order when shuffled),
    qzeros int32 [G,N/8], scales fp16 [G,N]; g_idx int [K] maps row->group
    (None: k // group_size with group_size = K/G).  The product of an integer
    in [-16,15] with an fp16 scale is exact in fp32.
    """
    qw = np.asarray(qweight).view(np.uint32)
    if shuffled and bits == 4:
        qw = unshuffle_4bit_word(qw)
    q = gptq_unpack(qw.view(np.int32), bits)               # [K,N]
    z = gptq_unpack_zeros(qzeros, bits)                    # [G,N]  (stored + 1, NOT masked: 8-bit 255 -> 256, as the
    #                                                        reference's uint32 `item + 1`, q_gemm.cu:1427)
    s = np.asarray(scales).astype(np.float32)              # [G,N]
    size_k = q.shape[0]
    groups = s.shape[0]
    if g_idx is None or len(g_idx) == 0:
        gs = group_size or (size_k // groups)
        g_idx = np.arange(size_k) // gs
    g_idx = np.asarray(g_idx).astype(np.int64)
    return (q - z[g_idx, :]).astype(np.float32) * s[g_idx, :]


def gptq_gemm(a, qweight, qzeros, scales, g_idx, use_exllama, bit=4):
    """ops.gptq_gemm semantics (q_gemm.cu:2238-2261), fp64 accumulate.

    use_exllama=True: qweight has been through gptq_shuffle; g_idx is then the
    *permutation* (argsort of the original g_idx, gptq.py:219-221) or empty,
    and the kernel gathers A[:, perm] (q_gemm.cu:219-226) while groups are
    k // group_size in the permuted order.
    use_exllama=False: qweight is in AutoGPTQ order and g_idx maps row->group.
    """
    assert bit in (2, 3, 4, 8)
    a = np.asarray(a).astype(np.float64)
    if use_exllama:
        w = gptq_dequant(qweight, qzeros, scales, None, shuffled=True, bits=bit)
        if g_idx is not None and len(g_idx) > 0:
            a = a[:, np.asarray(g_idx).astype(np.int64)]
    else:
        w = gptq_dequant(qweight, qzeros, scales, g_idx, shuffled=False, bits=bit)
    return a @ w.astype(np.float64)


# --------------------------------------------------------------------------
# AWQ tensors
# --------------------------------------------------------------------------
def awq_pack(q_w, num_bits=4):
    """quant_utils.py:423-441: interleave columns [0,2,4,6,1,3,5,7] then pack
    along N: nibble position p of word c holds column 8c + AWQ_ORDER[p]."""
    assert num_bits == 4
    q_w = np.asarray(q_w)
    size_k, size_n = q_w.shape
    t = q_w.reshape(-1, 8)[:, AWQ_ORDER].reshape(size_k, size_n)
    return pack_cols(t, num_bits)


def awq_unpack(packed):
    """Inverse of awq_pack -> int32 [K,N] (test_awq_triton.py:13-26)."""
    t = unpack_cols(packed, 4)
    k, n = t.shape
    return t.reshape(-1, 8)[:, AWQ_REVERSE_ORDER].reshape(k, n)


def awq_dequantize(qweight, scales, qzeros, group_size=None):
    """ops.awq_dequantize: W[K,N] fp16 = fp16((q - z) * s)
    (awq/gemm_kernels.cu:340-403: sub.f16x2 is exact, the multiply rounds
    once to fp16 == rounding the exact fp32 product)."""
    q = awq_unpack(qweight)
    z = awq_unpack(qzeros)
    s = np.asarray(scales).astype(np.float32)
    size_k = q.shape[0]
    gs = group_size or (size_k // s.shape[0])
    g = np.arange(size_k) // gs
    w = (q - z[g, :]).astype(np.float32) * s[g, :]
    return w.astype(np.float16)


def awq_gemm(a, qweight, scales, qzeros):
    """ops.awq_gemm semantics (awq/gemm_kernels.cu:784-841), fp64 accumulate.
    Positional order (in_feats, kernel, scaling_factors, zeros, split_k) --
    see SURVEY.md section 8b gotcha."""
    w = awq_dequantize(qweight, scales, qzeros).astype(np.float64)
    return np.asarray(a).astype(np.float64) @ w
