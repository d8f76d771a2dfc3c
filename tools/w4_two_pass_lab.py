#!/usr/bin/env python3
"""W4A16 prefill GEMM at M = 8192 (and 4096 / 2048): the fused eight-phase kernel (weights dequantised inside the K loop, once
per row-tile workgroup) against the two-pass form (dequantise-transpose once per call + the same schedule with the weights by
LDS-DMA; APHRO_WNA16_LARGE_TWO_PASS) and against dequantise + hipBLASLt.  Bits of the two forms must be identical."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aphrodite_engine_amd import _custom_ops as ops  # noqa: E402
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / n


Ms = [int(x) for x in sys.argv[1:]] or [8192]
for M in Ms:
    for K, N, nm in ((4096, 6144, "qkv"), (4096, 4096, "o"), (4096, 28672, "gate_up"), (14336, 4096, "down")):
        qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // 8, N), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
        qz = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // 128, N // 8), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
        sc = (torch.rand(K // 128, N, generator=g, device=dev) * 0.01 + 0.005).half()
        a = torch.randn(M, K, device=dev, dtype=torch.float16, generator=g)
        res = {}
        for name, val in (("fused", 0), ("two_pass", 1)):
            with ops.knob("APHRO_WNA16_LARGE_TWO_PASS", val):
                out = ops._wna16_large(a, qw, qz, sc, None, 1)
                res[name] = (out, timeit(lambda: ops._wna16_large(a, qw, qz, sc, None, 1)))
        same = torch.equal(res["fused"][0], res["two_pass"][0])
        with ops.knob("APHRO_WNA16_NO_LARGE", 1):
            empty = torch.empty(0, dtype=torch.int32, device=dev)
            t_lib = timeit(lambda: ops.gptq_gemm(a, qw, qz, sc, empty, True, 4))
        fl = 2.0 * M * K * N
        tf, tt = res["fused"][1], res["two_pass"][1]
        print(f"M={M} {nm:8s} {K}x{N}: fused {tf * 1e3:7.1f} us {fl / tf / 1e9:6.0f} TF | two-pass {tt * 1e3:7.1f} us {fl / tt / 1e9:6.0f} TF | "
              f"dequant + hipBLASLt {t_lib * 1e3:7.1f} us {fl / t_lib / 1e9:6.0f} TF | two-pass / library {t_lib / tt:5.3f} | bits equal: {same}")
        del qw, qz, sc, a, res
