#!/usr/bin/env python3
"""bench.py's prefill_section in a fresh process (no decode bench before it), three times back to back: is the 0.87 vs
1.06 PFLOP/s difference between the driver's bench line and tools/large_gemm_bench.py a clock / thermal effect of running
after the decode section, a first-call effect, or the measurement itself (3 calls, output allocated per call)?"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from aphrodite_engine_amd import model as M  # noqa: E402
from aphrodite_engine_amd import _custom_ops as ops  # noqa: E402

for rep in range(3):
    out = bench.prefill_section(M.LLAMA3_8B)
    print(json.dumps({k: round(v["TFLOPs"], 1) for k, v in out.items()}), flush=True)
# the W4A16 GEMM alone, output pre-allocated through the C entry point's workspace path, 10 calls
g = torch.Generator(device="cuda").manual_seed(7)
Mr, K, N = 8192, 4096, 28672
qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // 8, N), generator=g, device="cuda", dtype=torch.int64).to(torch.int32)
qz = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // 128, N // 8), generator=g, device="cuda", dtype=torch.int64).to(torch.int32)
sc = (torch.rand(K // 128, N, generator=g, device="cuda") * 0.01 + 0.005).half()
a = torch.randn(Mr, K, device="cuda", dtype=torch.float16, generator=g)
empty = torch.empty(0, dtype=torch.int32, device="cuda")
for iters in (3, 10, 30):
    ops.gptq_gemm(a, qw, qz, sc, empty, True, 4)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        ops.gptq_gemm(a, qw, qz, sc, empty, True, 4)
    e.record()
    e.synchronize()
    t = s.elapsed_time(e) * 1e-3 / iters
    print(json.dumps({"gptq_gemm 8192x4096x28672 iters": iters, "ms": round(t * 1e3, 3), "TFLOPs": round(2.0 * Mr * N * K / t / 1e12, 1)}), flush=True)
