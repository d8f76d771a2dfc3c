#!/usr/bin/env python3
"""What ONE resident copy of the int4 matrices costs the prompt-sized GEMMs (round 6): the tile machine on the strip-major
copy (ops.wna16_gemm_large_strip) against the same call on [K/8, N] (ops._wna16_large), the four Llama-3-8B shapes.
usage: python tools/one_copy_bench.py  (on a GPU box; prints one table -> profiles/r6_one_copy.txt)"""
import sys
import os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aphrodite_engine_amd import _custom_ops as ops  # noqa: E402

DEV = "cuda:0"


def timed(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    g = torch.Generator(device=DEV).manual_seed(0)
    print("W4A16 prompt-sized GEMM, [K/8, N] original vs the strip-major copy read in place (us per call, best of 3 x 20)")
    print(f"{'shape':>24} {'M':>6} {'rowmajor':>10} {'strip':>10} {'ratio':>7}")
    for name, K, N in (("qkv", 4096, 6144), ("o", 4096, 4096), ("gate_up", 4096, 28672), ("down", 14336, 4096)):
        qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // 8, N), generator=g, device=DEV, dtype=torch.int64).to(torch.int32)
        qz = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // 128, N // 8), generator=g, device=DEV, dtype=torch.int64).to(torch.int32)
        sc = (torch.rand(K // 128, N, generator=g, device=DEV) * 0.01 + 0.005).half()
        st = ops.wna16_strip_relayout(qw, 32, K // 128)
        for M in (256, 1024, 4096, 8192):
            a = (torch.randn(M, K, generator=g, device=DEV) * 0.5).half()
            t0 = min(timed(lambda: ops._wna16_large(a, qw, qz, sc, None, 1)) for _ in range(3))
            t1 = min(timed(lambda: ops.wna16_gemm_large_strip(a, st, qz, sc, 1)) for _ in range(3))
            print(f"{name + f' {K}x{N}':>24} {M:>6} {t0:>10.1f} {t1:>10.1f} {t1 / t0:>7.3f}")
        t2 = min(timed(lambda: ops.wna16_strip_unrelayout(st, 32, K // 128)) for _ in range(3))
        print(f"{'':>24} (the permutation backwards, for the plans that need [K/8, N]: {t2:.1f} us)")


if __name__ == "__main__":
    main()
