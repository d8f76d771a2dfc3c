"""W8A8 FP8 prefill-sized scaled GEMM (csrc/fp8_gemm_large.hip) vs torch._scaled_mm (hipBLASLt): correctness against
sa * (sb * (A @ B)) in fp32 on the decoded fp8 values, and MFMA throughput.   python tools/fp8_large_bench.py [M ...]"""
import os
import sys

import torch

from aphrodite_engine_amd import _custom_ops as ops

DEV = "cuda"
SHAPES = [(4096, 6144), (4096, 4096), (4096, 28672), (14336, 4096)]
Ms = [int(x) for x in sys.argv[1:]] or [128, 256, 1024, 2048, 8192]


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) * 1e-3 / iters


def lib_mm(a, bt, sa, sb):
    os.environ["APHRO_FP8_NO_LARGE"] = "1"
    try:
        return ops.cutlass_scaled_mm(a, bt, sa, sb, torch.bfloat16)
    finally:
        del os.environ["APHRO_FP8_NO_LARGE"]


g = torch.Generator(device=DEV).manual_seed(0)
for K, N in SHAPES:
    w = (torch.randn(N, K, generator=g, device=DEV) * 0.5).to(torch.float8_e4m3fn)
    sb = torch.rand(N, generator=g, device=DEV) * 0.01 + 0.005
    for M in Ms:
        a = torch.randn(M, K, generator=g, device=DEV).to(torch.float8_e4m3fn)
        sa = torch.rand(M, 1, generator=g, device=DEV) * 0.1 + 0.05
        ref = sa * (sb.view(1, N) * (a.float() @ w.float().t()))
        got = ops.cutlass_scaled_mm(a, w.t(), sa, sb, torch.bfloat16)
        err = (got.float() - ref).abs().max().item() / ref.abs().max().item()
        t_new = timeit(lambda: ops.cutlass_scaled_mm(a, w.t(), sa, sb, torch.bfloat16))
        t_lib = timeit(lambda: lib_mm(a, w.t(), sa, sb))
        fl = 2.0 * M * N * K
        print(f"K={K:5d} N={N:5d} M={M:5d}: large kernel {t_new * 1e6:9.1f} us {fl / t_new / 1e12:7.1f} TF | "
              f"torch._scaled_mm {t_lib * 1e6:9.1f} us {fl / t_lib / 1e12:7.1f} TF | max rel err {err:.2e}", flush=True)
