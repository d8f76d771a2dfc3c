#!/usr/bin/env python3
"""Runs the prefill flash-attention kernel a few times (for rocprofv3 --pmc / --kernel-trace)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aphrodite_engine_amd import _custom_ops as ops  # noqa: E402
T = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
Hq, Hkv, D = 32, 8, 128
qkv = torch.randn(T, (Hq + 2 * Hkv) * D, device="cuda", dtype=torch.float16) * 0.5
q = qkv[:, :Hq * D].view(T, Hq, D)
k = qkv[:, Hq * D:(Hq + Hkv) * D].view(T, Hkv, D)
v = qkv[:, (Hq + Hkv) * D:].view(T, Hkv, D)
cu = torch.tensor([0, T], dtype=torch.int32, device="cuda")
for _ in range(4):
    ops.flash_attn_varlen(q, k, v, cu, T, D ** -0.5, causal=True)
torch.cuda.synchronize()
