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
