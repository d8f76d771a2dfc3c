#!/usr/bin/env python3
"""Time of one fused sampling launch at the decode shape (32 x 128256 f16), by filter configuration."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aphrodite_engine_amd import _custom_ops as ops  # noqa: E402

dev = "cuda"
b, v = 32, 128256
logits = (torch.randn(b, v, device=dev) * 1.2).half()
seeds = torch.arange(b, dtype=torch.int64, device=dev)
q = torch.empty(b, v, device=dev).exponential_()
cases = {
    "no filters, q given": dict(q=q),
    "no filters, in-kernel noise": dict(seeds=seeds),
    "T 0.8": dict(temperature=torch.full((b, ), 0.8), seeds=seeds),
    "top-k 50": dict(top_k=torch.full((b, ), 50, dtype=torch.int32), seeds=seeds),
    "top-p 0.95": dict(top_p=torch.full((b, ), 0.95), seeds=seeds),
    "T 0.8 + top-k 50 + top-p 0.95": dict(temperature=torch.full((b, ), 0.8), top_k=torch.full((b, ), 50, dtype=torch.int32),
                                         top_p=torch.full((b, ), 0.95), seeds=seeds),
}
for name, kw in cases.items():
    kw = {k: (t.to(dev) if isinstance(t, torch.Tensor) else t) for k, t in kw.items()}
    for _ in range(3):
        ops.sample_top_k_top_p(logits, **kw)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        ops.sample_top_k_top_p(logits, **kw)
    e.record()
    e.synchronize()
    print(f"{name:36s} {s.elapsed_time(e) / 20 * 1e3:8.1f} us")
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20):
    ops.argmax_rows(logits)
e.record()
e.synchronize()
print(f"{'argmax_rows (greedy)':36s} {s.elapsed_time(e) / 20 * 1e3:8.1f} us")
