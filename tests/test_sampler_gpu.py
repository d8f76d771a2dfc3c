"""Fused sampling kernel vs the sort-based restatement of the reference sampler (oracle/sampling.py):
same surviving set (checked through the sample and through forced-choice probes), same sample given the
same Exp(1) draws; in-kernel noise checked statistically."""
import numpy as np
import pytest
import torch

from oracle import sampling as osamp

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from aphrodite_engine_amd import _custom_ops
    return _custom_ops


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("v", [1000, 1003, 32000, 128256])      # 1003: rows not 16-byte aligned (scalar loads)
def test_sample_matches_reference_restatement(ops, dtype, v):
    rng = np.random.default_rng(v)
    b = 12
    logits = torch.from_numpy((rng.standard_normal((b, v)) * 3).astype(np.float32)).to(dtype)
    temperature = np.array([1.0, 0.7, 1.3, 0.0, 2.0, 0.5, 1.0, 1.0, 0.9, 1.1, 1.0, 0.3], np.float32)
    top_k = np.array([50, 0, 1, 40, v, 7, 0, 1000, -1, 3, 64, 0], np.int32)
    top_p = np.array([0.9, 0.8, 1.0, 0.95, 0.5, 1.0, 1.0, 0.3, 0.99, 0.6, 0.0, 0.05], np.float32)
    q = rng.exponential(size=(b, v)).astype(np.float32)
    want, masked = osamp.sample(logits.float().numpy(), temperature, top_k, top_p, q)
    padded = torch.zeros(b, v + 24, dtype=dtype)            # strided rows
    padded[:, :v] = logits
    got = ops.sample_top_k_top_p(padded.to(DEV)[:, :v], torch.from_numpy(temperature), torch.from_numpy(top_k),
                                 torch.from_numpy(top_p), torch.from_numpy(q).to(DEV))
    got = got.cpu().numpy()
    kept = np.isfinite(masked)
    for r in range(b):   # 16-bit logits have ties: the sample must at least lie in the reference's kept set
        assert kept[r, got[r]], f"row {r}: token {got[r]} is outside the reference's top-k/top-p set"
    if dtype == torch.float32:
        np.testing.assert_array_equal(got, want)
    else:
        assert (got == want).mean() >= 0.9
    # the kept set itself, probed: noise that makes exactly one candidate win.  A token the reference
    # masked must never be sampled even when its draw is overwhelmingly favourable
    r = 0
    order = np.argsort(-logits[r].float().numpy() / max(temperature[r], 1e-5))
    inside, outside = order[int(kept[r].sum()) - 1], order[int(kept[r].sum())]
    if dtype == torch.float32:
        for tok, expect_in in ((inside, True), (outside, False)):
            qq = np.full((1, v), 1e6, np.float32)
            qq[0, tok] = 1e-30
            g = ops.sample_top_k_top_p(logits[r:r + 1].to(DEV), torch.from_numpy(temperature[r:r + 1]),
                                       torch.from_numpy(top_k[r:r + 1]), torch.from_numpy(top_p[r:r + 1]),
                                       torch.from_numpy(qq).to(DEV)).item()
            assert (g == tok) == expect_in


def test_sample_min_p(ops):
    rng = np.random.default_rng(9)
    b, v = 8, 32000
    logits = (rng.standard_normal((b, v)) * 3).astype(np.float32)
    temperature = np.array([1.0, 0.7, 1.3, 1.0, 2.0, 0.5, 1.0, 1.0], np.float32)
    top_k = np.array([0, 50, 0, 40, 0, 0, 200, 0], np.int32)
    top_p = np.array([1.0, 1.0, 0.9, 0.95, 1.0, 0.8, 1.0, 1.0], np.float32)
    min_p = np.array([0.05, 0.1, 0.02, 0.3, 0.01, 0.0, 0.9, 1.0], np.float32)
    q = rng.exponential(size=(b, v)).astype(np.float32)
    want, masked = osamp.sample(logits, temperature, top_k, top_p, q, min_p=min_p)
    got = ops.sample_top_k_top_p(torch.from_numpy(logits).to(DEV), torch.from_numpy(temperature),
                                 torch.from_numpy(top_k), torch.from_numpy(top_p), torch.from_numpy(q).to(DEV),
                                 min_p=torch.from_numpy(min_p)).cpu().numpy()
    assert np.isfinite(masked)[np.arange(b), got].all()
    np.testing.assert_array_equal(got, want)
    # logprob of the sampled token: log_softmax of the masked row (sampler.py:545)
    lp = torch.empty(b, dtype=torch.float32, device=DEV)
    got2 = ops.sample_top_k_top_p(torch.from_numpy(logits).to(DEV), torch.from_numpy(temperature),
                                  torch.from_numpy(top_k), torch.from_numpy(top_p), torch.from_numpy(q).to(DEV),
                                  min_p=torch.from_numpy(min_p), logprobs_out=lp).cpu().numpy()
    np.testing.assert_array_equal(got2, want)
    x = masked.astype(np.float64)          # after temperature, top-k, top-p, min-p (-inf = masked)
    ref_lp = x[np.arange(b), want] - (np.log(np.exp(x - x.max(1, keepdims=True)).sum(1)) + x.max(1))
    np.testing.assert_allclose(lp.cpu().numpy(), ref_lp, atol=2e-5, rtol=1e-5)


def test_sample_disabled_filters_and_errors(ops):
    rng = np.random.default_rng(3)
    logits = torch.from_numpy(rng.standard_normal((4, 5000)).astype(np.float32))
    q = rng.exponential(size=(4, 5000)).astype(np.float32)
    want = osamp.multinomial(osamp.softmax32(logits.numpy()), q)
    got = ops.sample_top_k_top_p(logits.to(DEV), q=torch.from_numpy(q).to(DEV))
    np.testing.assert_array_equal(got.cpu().numpy(), want)
    with pytest.raises(RuntimeError, match="noise q or per-row seeds"):
        ops.sample_top_k_top_p(logits.to(DEV))
    with pytest.raises(RuntimeError, match="one entry per row"):
        ops.sample_top_k_top_p(logits.to(DEV), temperature=torch.ones(3), q=torch.from_numpy(q).to(DEV))
    with pytest.raises(RuntimeError):
        ops.sample_top_k_top_p(logits, q=torch.from_numpy(q))          # CPU tensors: no fallback


def test_sample_in_kernel_noise_follows_the_distribution(ops):
    """seeds instead of q: 40 000 rows of the same 12-token distribution -> chi-square against softmax
    restricted to the top-p set; deterministic for equal seeds, different for different seeds."""
    rows, v = 40000, 12
    base = torch.tensor([2.0, 1.5, 1.0, 0.5, 0.0, -0.5, -1.0, -1.5, -2.0, -2.5, -3.0, -8.0])
    logits = base.repeat(rows, 1).to(DEV)
    seeds = torch.arange(rows, dtype=torch.int64) * 7919 + 13
    top_p = torch.full((rows, ), 0.9)
    a = ops.sample_top_k_top_p(logits, top_p=top_p, seeds=seeds)
    b = ops.sample_top_k_top_p(logits, top_p=top_p, seeds=seeds)
    c = ops.sample_top_k_top_p(logits, top_p=top_p, seeds=seeds + 1)
    assert torch.equal(a, b) and not torch.equal(a, c)
    masked = osamp.apply_top_k_top_p(base.numpy()[None, :], np.array([0.9], np.float32), np.array([v]))
    probs = osamp.softmax32(masked)[0].astype(np.float64)
    counts = np.bincount(a.cpu().numpy(), minlength=v).astype(np.float64)
    assert counts[probs == 0].sum() == 0
    live = probs > 0
    chi2 = (((counts[live] - rows * probs[live]) ** 2) / (rows * probs[live])).sum()
    assert chi2 < 40.0, chi2        # ~ dof 7-8: the 1e-6 quantile is ~45
