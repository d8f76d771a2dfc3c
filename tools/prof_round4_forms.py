#!/usr/bin/env python3
"""Launches the round-4 forms a few times on cold weights (for `rocprofv3 --pmc FETCH_SIZE`, own pass):
  * gate_up 4096 x 28672 at 32 rows: the stream kernel (reference point) and the norm-in-consumer launch;
  * gate_up 4096 x 28672 at 64 rows on two 32-row halves (are the weights fetched from HBM once or twice?);
  * the 8192 x 7168 gate_up of a Llama-3-70B TP-8 shard at 64 rows on two halves (K-sliced 4 ways)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aphrodite_engine_amd import _custom_ops as ops  # noqa: E402
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)


def gptq(k, n):
    qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (k // 8, n), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
    qz = torch.randint(-2 ** 31, 2 ** 31 - 1, (k // 128, n // 8), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
    sc = (torch.rand(k // 128, n, generator=g, device=dev) * 0.01).half()
    return qw, qz, sc


NC = 6
K, N = 4096, 28672
ws = [gptq(K, N) for _ in range(NC)]
strips = [ops.wna16_strip_relayout(w[0], 32, K // 128) for w in ws]
a32 = ops.wna16_pack_a(torch.randn(32, K, device=dev, dtype=torch.float16))
a64 = ops.wna16_pack_a(torch.randn(64, K, device=dev, dtype=torch.float16))
slabs = torch.randn(4, 32, K, device=dev) * 0.25
res = torch.randn(32, K, device=dev).half()
w = torch.ones(K, device=dev).half()
sync = torch.zeros(NC, dtype=torch.int32, device=dev)
for i, ((qw, qz, sc), st) in enumerate(zip(ws, strips)):
    ops.wna16_gemm_resident(a32, 32, K, st, qz, sc, 1, mode="silu", strip_layout=True)            # stream kernel, 32 rows
    ops.wna16_gemm_norm_fused(slabs, res, w, 1e-5, st, qz, sc, 1, sync[i:i + 1], mode="silu")     # norm-in-consumer
    ops.wna16_gemm_resident(a64, 64, K, st, qz, sc, 1, mode="silu", strip_layout=True)            # two 32-row halves
torch.cuda.synchronize()
del ws, strips
K2, N2 = 8192, 7168
ws2 = [gptq(K2, N2) for _ in range(NC)]
a64b = ops.wna16_pack_a(torch.randn(64, K2, device=dev, dtype=torch.float16))
for qw, qz, sc in ws2:
    st = ops.wna16_strip_relayout(qw, 64, K2 // 128)
    ops.wna16_gemm_resident(a64b, 64, K2, st, qz, sc, 1, mode="slabs", strip_layout=True)         # shard gate_up, halves, 4 K slices
torch.cuda.synchronize()
