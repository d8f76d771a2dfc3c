"""W4A16 GEMM in the mid-M regime (decode batches of 48..512 rows): the decode kernel (64-row blocks, csrc/wna16_gemm.hip),
the prefill-sized kernel (csrc/wna16_gemm_large.hip) and the HBM / MFMA bounds of each shape.
    python tools/mid_gemm_bench.py [M ...]"""
import os
import sys

import torch

from aphrodite_engine_amd import _custom_ops as ops

DEV = "cuda"
SHAPES = [(4096, 6144), (4096, 4096), (4096, 28672), (14336, 4096)]
Ms = [int(x) for x in sys.argv[1:]] or [32, 48, 64, 96, 128, 192, 256, 512]


def timeit(fn, iters=20, reps=5):
    """HIP-graph replay of `iters` back-to-back calls: no host launch overhead in the number."""
    fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(iters):
            fn()
    gr.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        gr.replay()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) * 1e-3 / (iters * reps)


def small(a, qw, qz, sc):
    os.environ["APHRO_WNA16_NO_LARGE"] = "1"
    os.environ["APHRO_WNA16_NO_MID"] = "1"
    try:
        return ops._wna16(a, qw, qz, sc, None, 1)
    finally:
        del os.environ["APHRO_WNA16_NO_LARGE"], os.environ["APHRO_WNA16_NO_MID"]


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
        t_small = timeit(lambda: small(a, qw, qz, sc))
        got = ops._wna16_large(a, qw, qz, sc, None, 1)
        err = (got.float() - ref).abs().max().item() / ref.abs().max().item()
        t_large = timeit(lambda: ops._wna16_large(a, qw, qz, sc, None, 1))
        t_mid = None
        if ops.wna16_mid_ok(M, N, K, G):
            gm = ops._wna16_mid(a, qw, qz, sc, None, 1)
            err_mid = (gm.float() - ref).abs().max().item() / ref.abs().max().item()
            t_mid = timeit(lambda: ops._wna16_mid(a, qw, qz, sc, None, 1))
        by = K * N / 2 + G * N * 2.5 + M * K * 2 + M * N * 2
        fl = 2.0 * M * N * K
        bound = max(by / 8e12, fl / 2.5e15)
        mid = f" | mid kernel {t_mid * 1e6:7.1f} us (err {err_mid:.1e})" if t_mid is not None else ""
        print(f"K={K:5d} N={N:5d} M={M:4d}: decode kernel {t_small * 1e6:7.1f} us | large kernel {t_large * 1e6:7.1f} us "
              f"(err {err:.1e}){mid} | bound {bound * 1e6:5.1f} us ({'hbm' if by / 8e12 > fl / 2.5e15 else 'mfma'}) "
              f"| best/bound {bound / min(t_small, t_large, t_mid or 1.0):.2f}", flush=True)
