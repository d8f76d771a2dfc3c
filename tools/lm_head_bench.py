#!/usr/bin/env python3
"""LM head of Llama-3 ([128256, 4096] f16) at decode batch sizes: the one-launch GEMM + argmax (csrc/lm_head.hip) against
hipBLASLt (torch.matmul) + argmax_rows.  The 1.05 GB of weights exceed the 256 MiB Infinity Cache, so every call streams
from HBM; HIP-graph replay, HIP-event timing."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aphrodite_engine_amd import _custom_ops as ops  # noqa: E402
from tools.prefetch_lab import timeit  # noqa: E402


def main():
    V, K = 128256, 4096
    g = torch.Generator(device="cuda").manual_seed(0)
    w = (torch.randn(V, K, device="cuda", generator=g) * 0.02).half()
    rows = []
    for M in (32, 16, 8, 1):
        h = (torch.randn(M, K, device="cuda", generator=g) * 0.5).half()
        out = torch.empty(M, dtype=torch.int64, device="cuda")
        logits = torch.empty(M, V, dtype=torch.float16, device="cuda")
        nb = V * K * 2

        def lib():
            lg = torch.matmul(h, w.t())
            ops.argmax_rows(lg, out)

        def fused():
            ops.lm_head_argmax(h, w, V, out=out)

        def fused_logits():
            ops.lm_head_argmax(h, w, V, out=out, logits_out=logits)
        for name, fn in (("hipBLASLt + argmax_rows", lib), ("lm_head_argmax", fused), ("lm_head_argmax + logits", fused_logits)):
            tt = timeit(fn, 1)
            r = dict(M=M, impl=name, us=round(tt * 1e6, 1), TBps=round(nb / tt / 1e12, 3))
            rows.append(r)
            print(json.dumps(r), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "lm_head_bench.jsonl"), "w") as f:
        for r in rows:
            f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
