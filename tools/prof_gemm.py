#!/usr/bin/env python3
"""Runs ONE hot kernel a few times (for rocprofv3 --pmc / --kernel-trace)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aphrodite_engine_amd import _custom_ops as ops  # noqa: E402
which = sys.argv[1] if len(sys.argv) > 1 else "gate_up"
M = int(sys.argv[2]) if len(sys.argv) > 2 else 32
SH = {"qkv": (4096, 6144), "o": (4096, 4096), "gate_up": (4096, 28672), "down": (14336, 4096)}
K, N = SH[which]
G = K // 128
g = torch.Generator(device="cuda").manual_seed(0)
ws = []
for _ in range(10):
    qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // 8, N), generator=g, device="cuda", dtype=torch.int64).to(torch.int32)
    qz = torch.randint(-2 ** 31, 2 ** 31 - 1, (G, N // 8), generator=g, device="cuda", dtype=torch.int64).to(torch.int32)
    sc = (torch.rand(G, N, generator=g, device="cuda") * 0.01).half()
    ws.append((qw, qz, sc))
a = torch.randn(M, K, device="cuda", dtype=torch.float16)
e = torch.empty(0, dtype=torch.int32, device="cuda")
for _ in range(3):
    for qw, qz, sc in ws:
        ops.gptq_gemm(a, qw, qz, sc, e, True, 4)
torch.cuda.synchronize()
