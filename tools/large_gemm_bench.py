"""W4A16 prefill-sized GEMM (csrc/wna16_gemm_large.hip) vs the dequant + library GEMM it replaces: correctness against
x @ dequant(W) in fp32 and MFMA throughput.   python tools/large_gemm_bench.py [M ...]"""
import sys
import time

import torch

from aphrodite_engine_amd import _custom_ops as ops

DEV = "cuda"
SHAPES = [(4096, 6144), (4096, 4096), (4096, 28672), (14336, 4096)]
Ms = [int(x) for x in sys.argv[1:]] or [128, 256, 2048, 8192]


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


g = torch.Generator(device=DEV).manual_seed(0)
for K, N in SHAPES:
    G = K // 128
    qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // 8, N), generator=g, device=DEV, dtype=torch.int64).to(torch.int32)
    qz = torch.randint(-2 ** 31, 2 ** 31 - 1, (G, N // 8), generator=g, device=DEV, dtype=torch.int64).to(torch.int32)
    sc = (torch.rand(G, N, generator=g, device=DEV) * 0.01 + 0.005).half()
    w = ops.gptq_dequant(qw, qz, sc, None, True)
    for M in Ms:
        a = torch.randn(M, K, generator=g, device=DEV).half()
        ref = a.float() @ w.float()
        got = ops._wna16_large(a, qw, qz, sc, None, 1)
        err = (got.float() - ref).abs().max().item() / ref.abs().max().item()
        t_new = timeit(lambda: ops._wna16_large(a, qw, qz, sc, None, 1))
        t_lib = timeit(lambda: torch.matmul(a, ops.gptq_dequant(qw, qz, sc, None, True)))
        t_mm = timeit(lambda: torch.matmul(a, w))
        fl = 2.0 * M * N * K
        print(f"K={K:5d} N={N:5d} M={M:5d}: large kernel {t_new * 1e6:9.1f} us {fl / t_new / 1e12:7.1f} TF | dequant+hipBLASLt "
              f"{t_lib * 1e6:9.1f} us {fl / t_lib / 1e12:7.1f} TF | hipBLASLt alone {fl / t_mm / 1e12:7.1f} TF | max rel err {err:.2e}",
              flush=True)
