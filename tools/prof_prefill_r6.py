#!/usr/bin/env python3
"""Launches the round-5/6 prefill kernels -- flash_attn_varlen_v4_kernel (T = 8192 causal, Hq 32 / Hkv 8),
wna16_gemm_large8_kernel and fp8_gemm_large8_kernel at the four Llama-3-8B projection shapes, M = 8192 -- NL times each, for
separate rocprofv3 passes:

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -- python tools/prof_prefill_r6.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace ...
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
              SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace ...

reduced by tools/pmc_prefill.py into profiles/r6_pmc_prefill.{txt,json}.  A 512 MB fill between launches evicts the
Infinity Cache (256 MiB), so FETCH_SIZE counts what a prompt's layer sees: operands that were last touched a layer ago."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aphrodite_engine_amd import _custom_ops as ops  # noqa: E402

NL = int(os.environ.get("PROF_NL", "3"))
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)


def evict():
    flush.fill_(1)


T, Hq, Hkv, D = 8192, 32, 8, 128
qkv = torch.randn(T, (Hq + 2 * Hkv) * D, device=dev, dtype=torch.float16, generator=g) * 0.5
q, k, v = qkv[:, :Hq * D].view(T, Hq, D), qkv[:, Hq * D:(Hq + Hkv) * D].view(T, Hkv, D), qkv[:, (Hq + Hkv) * D:].view(T, Hkv, D)
cu = torch.tensor([0, T], dtype=torch.int32, device=dev)
for _ in range(NL + 1):
    evict()
    ops.flash_attn_varlen(q, k, v, cu, T, D ** -0.5, causal=True)

M = 8192
empty = torch.empty(0, dtype=torch.int32, device=dev)
for K, N in ((4096, 6144), (4096, 4096), (4096, 28672), (14336, 4096)):
    qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // 8, N), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
    qz = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // 128, N // 8), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
    sc = (torch.rand(K // 128, N, generator=g, device=dev) * 0.01 + 0.005).half()
    a = torch.randn(M, K, device=dev, dtype=torch.float16, generator=g)
    for _ in range(NL + 1):
        evict()
        ops.gptq_gemm(a, qw, qz, sc, empty, True, 4)
    w8 = (torch.randn(N, K, device=dev, generator=g) * 0.5).to(torch.float8_e4m3fn)
    a8 = a.to(torch.float8_e4m3fn)
    sa = torch.rand(M, 1, device=dev, generator=g) * 0.1 + 0.05
    sb = torch.rand(N, device=dev, generator=g) * 0.01 + 0.005
    ob = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(NL + 1):
        evict()
        ops.cutlass_scaled_mm(a8, w8.t(), sa, sb, torch.bfloat16, out=ob)
    del qw, qz, sc, a, w8, a8, sa, sb, ob
torch.cuda.synchronize()
