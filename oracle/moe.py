"""CPU restatement of the mixture-of-experts routing ops and of the reference's MoE layer.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PINNED by tests/golden/moe.npz (the reference's
torch_moe run in the build container) and by the worked example in fused_moe.py:199-212.

  topk_softmax            kernels/moe/softmax.cu:17-520 (softmax in fp32, k rounds of arg-max,
                          ties -> lowest index; weights not renormalised; source row k*T + t)
  moe_align_block_size    kernels/moe/align_block_size_kernel.cu:17-126 + the docstring example of
                          fused_moe.py:199-212 (stable by slot index, padding value = numel)
  moe_layer               tests/kernels/test_moe.py torch_moe (:19-35) / mixtral_quant.py:130-156:
                          dense per-expert MLP, routed weights, sum
"""
import numpy as np


def topk_softmax(gating, topk):
    g = np.asarray(gating, dtype=np.float32)
    t, e = g.shape
    mx = g.max(axis=1, keepdims=True)
    ex = np.exp((g - mx).astype(np.float32)).astype(np.float32)
    p = (ex * (np.float32(1.0) / ex.sum(axis=1, keepdims=True, dtype=np.float32))).astype(np.float32)
    w = np.zeros((t, topk), np.float32)
    ids = np.zeros((t, topk), np.int32)
    src = np.zeros((t, topk), np.int32)
    for i in range(t):
        row = p[i].copy()
        for k in range(topk):
            j = int(np.argmax(row))           # first maximum = lowest index
            w[i, k], ids[i, k], src[i, k] = p[i, j], j, k * t + i
            row[j] = -1.0
    return w, ids, src


def moe_align_block_size(topk_ids, num_experts, block_size):
    flat = np.asarray(topk_ids).reshape(-1)
    numel = flat.size
    max_padded = numel + num_experts * (block_size - 1)
    sorted_ids = np.full(max_padded, numel, np.int32)
    expert_ids = np.full((max_padded + block_size - 1) // block_size, -1, np.int32)
    pos = 0
    for e in range(num_experts):
        idx = np.nonzero(flat == e)[0]
        sorted_ids[pos:pos + idx.size] = idx
        nblk = (idx.size + block_size - 1) // block_size
        expert_ids[pos // block_size:pos // block_size + nblk] = e
        pos += nblk * block_size
    return sorted_ids, expert_ids, pos


def moe_layer(x, w13, w2, topk_weights, topk_ids):
    """x [T,H] float; w13 [E,H,2I] (gate | up), w2 [E,I,H] dequantised float weights."""
    x = np.asarray(x, np.float64)
    t = x.shape[0]
    out = np.zeros((t, w2.shape[2]), np.float64)
    inter = w2.shape[1]
    for i in range(t):
        for k in range(topk_ids.shape[1]):
            e = int(topk_ids[i, k])
            h = x[i] @ np.asarray(w13[e], np.float64)
            g, u = h[:inter], h[inter:]
            a = g / (1.0 + np.exp(-g)) * u
            out[i] += float(topk_weights[i, k]) * (a @ np.asarray(w2[e], np.float64))
    return out


def fused_experts_fp8(x, w13_bits, w2_bits, w13_scale, w2_scale, topk_weights, topk_ids, a1_scale=None, a2_scale=None,
                      dtype="float16"):
    """fused_experts(use_fp8_w8a8=True), aphrodite/modeling/layers/fused_moe/fused_moe.py:566-690 with the kernel
    epilogue of :150-163, restated with the reference's tensor layouts and roundings:

      A1_q, s1 = scaled_fp8_quant(x, a1_scale)            per tensor (dynamic: absmax over the whole tensor)
      cache1[t, j] = T(acc * s1 * w13_scale[e])            acc = A1_q[t] . W13_q[e]^T (exact products, fp64 sums here)
      cache2 = silu_and_mul(cache1)                        in T (activation_kernels.cu:14-28: fp32 math, one rounding)
      A2_q, s2 = scaled_fp8_quant(cache2, a2_scale)
      cache3[t, j] = T((acc * w[t, j]) * s2 * w2_scale[e])
      out[t] = T(sum_j cache3[t, j])                        torch.sum: fp32 accumulate, one rounding

    x [M, H] in T; w13_bits [E, 2I, H] / w2_bits [E, H, I] uint8 e4m3; scales float32.  Returns (out, cache1, cache3)."""
    import torch
    from . import fp8 as ofp8
    tdt = getattr(torch, dtype)

    def rnd(a):        # one rounding of fp32 values to T, back as float32
        return torch.from_numpy(np.asarray(a, np.float32)).to(tdt).float().numpy()
    x = np.asarray(x, np.float32)
    m, h = x.shape
    k = topk_ids.shape[1]
    inter = w2_bits.shape[2]
    a1q, s1 = ofp8.scaled_fp8_quant(x, a1_scale)
    s1 = np.float32(np.asarray(s1).reshape(-1)[0])
    a1 = ofp8.fp8_decode(a1q, "e4m3").astype(np.float64)
    w13 = ofp8.fp8_decode(np.asarray(w13_bits), "e4m3").astype(np.float64)
    w2 = ofp8.fp8_decode(np.asarray(w2_bits), "e4m3").astype(np.float64)
    cache1 = np.zeros((m * k, 2 * inter), np.float32)
    for t in range(m):
        for j in range(k):
            e = int(topk_ids[t, j])
            acc = (a1[t] @ w13[e].T).astype(np.float32)
            cache1[t * k + j] = rnd((acc * s1) * np.float32(w13_scale[e]))
    g, u = cache1[:, :inter], cache1[:, inter:]
    cache2 = rnd((g / (np.float32(1.0) + np.exp(-g, dtype=np.float32))).astype(np.float32) * u)
    a2q, s2 = ofp8.scaled_fp8_quant(cache2, a2_scale)
    s2 = np.float32(np.asarray(s2).reshape(-1)[0])
    a2 = ofp8.fp8_decode(a2q, "e4m3").astype(np.float64)
    cache3 = np.zeros((m * k, h), np.float32)
    for t in range(m):
        for j in range(k):
            e = int(topk_ids[t, j])
            acc = (a2[t * k + j] @ w2[e].T).astype(np.float32)
            cache3[t * k + j] = rnd(((acc * np.float32(topk_weights[t, j])) * s2) * np.float32(w2_scale[e]))
    out = rnd(cache3.reshape(m, k, h).sum(axis=1, dtype=np.float32))
    return out, cache1, cache3
