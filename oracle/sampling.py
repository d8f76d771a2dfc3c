"""CPU restatement of the reference's random sampling path -- TEST INFRASTRUCTURE ONLY (the product
never imports this).  Follows aphrodite/modeling/layers/sampler.py:
  temperature      logits.div_(t), t < 1e-5 -> 1.0        (:256-262, sampling_metadata.py:459-461)
  _apply_top_k_top_p  ascending sort, k-th largest as threshold (`<` masks, ties at the threshold
                   stay), softmax of the masked row, ascending cumsum <= 1 - p masks, last stays  (:865-891)
  _apply_min_p     probs < min_p * max(probs) masked                                             (:894-908)
  _multinomial     q ~ Exp(1); argmax(probs / q)                                                 (:1273-1292)
all in float32 like the reference (logits are cast to float first, :232).  PINNED: the reference
holds no golden vectors for the sampler, so tests/golden/make_golden.py lifts `_apply_top_k_top_p` and
`_multinomial` out of the reference file with ``ast`` and runs them here; tests/golden/sampler.npz
holds their outputs and tests/test_oracle_golden.py::test_sampler_golden checks this restatement
against them (masked logits exact, sampled ids exact)."""
import numpy as np


def apply_temperature(logits, temperature):
    t = np.asarray(temperature, np.float32).copy()
    t[t < 1e-5] = 1.0
    return (np.asarray(logits, np.float32) / t[:, None]).astype(np.float32)


def softmax32(x):
    m = x.max(axis=-1, keepdims=True)
    e = np.exp((x - m).astype(np.float32)).astype(np.float32)
    return (e / e.sum(axis=-1, keepdims=True, dtype=np.float32)).astype(np.float32)


def apply_top_k_top_p(logits, p, k):
    """sampler.py:865-891 on float32 [B, V]; k int [B] (V = disabled), p float [B] (1.0 = disabled)."""
    logits = np.asarray(logits, np.float32)
    b, v = logits.shape
    idx = np.argsort(logits, axis=-1, kind="stable")
    srt = np.take_along_axis(logits, idx, axis=-1)
    kth = np.take_along_axis(srt, (v - np.asarray(k, np.int64))[:, None], axis=-1)
    srt = np.where(srt < kth, -np.inf, srt).astype(np.float32)
    probs = softmax32(srt)
    csum = np.cumsum(probs, axis=-1, dtype=np.float32)
    mask = csum <= (np.float32(1.0) - np.asarray(p, np.float32))[:, None]
    mask[:, -1] = False
    srt = np.where(mask, -np.inf, srt).astype(np.float32)
    out = np.empty_like(srt)
    np.put_along_axis(out, idx, srt, axis=-1)
    return out


def apply_min_p(logits, min_p):
    """sampler.py:894-908: tokens whose probability is below min_p * (largest probability) are masked."""
    logits = np.asarray(logits, np.float32).copy()
    probs = softmax32(logits)
    top = probs.max(axis=-1, keepdims=True)
    scaled = (np.asarray(min_p, np.float32)[:, None] * top).astype(np.float32)
    logits[probs < scaled] = -np.inf
    return logits


def multinomial(probs, q):
    """argmax(probs / q) per row, q ~ Exp(1) drawn by the caller (sampler.py:1273-1292)."""
    return np.argmax((np.asarray(probs, np.float32) / np.asarray(q, np.float32)).astype(np.float32), axis=-1)


def sample(logits, temperature, top_k, top_p, q, min_p=None):
    x = apply_temperature(logits, temperature)
    v = x.shape[1]
    k = np.asarray(top_k, np.int64).copy()
    k[(k <= 0) | (k > v)] = v
    x = apply_top_k_top_p(x, top_p, k)
    if min_p is not None:
        x = apply_min_p(x, min_p)
    return multinomial(softmax32(x), q), x
