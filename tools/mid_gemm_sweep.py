"""Sweep of the mid-M forms of csrc/wna16_gemm_large.hip (m-blocks per wave, column waves, K slices) at decode batch
sizes; lab knobs APHRO_WNA16_LARGE_{MB,WN,KSPLIT} are read per call.   python tools/mid_gemm_sweep.py [M ...]"""
import os
import sys

import torch

from aphrodite_engine_amd import _custom_ops as ops

DEV = "cuda"
SHAPES = [(4096, 28672), (14336, 4096), (4096, 6144), (4096, 4096)]
Ms = [int(x) for x in sys.argv[1:]] or [32, 64, 128]
KS = [int(x) for x in os.environ.get("SWEEP_KS", "1,2,4,8,16").split(",")]


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
        mbs = [1] if M <= 32 else [2] if M <= 64 else [4]
        row = []
        for mb in mbs:
            for wn in (2, 4):
                for ks in KS:
                    os.environ.update(APHRO_WNA16_LARGE_MB=str(mb), APHRO_WNA16_LARGE_WN=str(wn), APHRO_WNA16_LARGE_KSPLIT=str(ks))
                    try:
                        got = ops._wna16_large(a, qw, qz, sc, None, 1)
                        err = (got.float() - ref).abs().max().item() / ref.abs().max().item()
                        t = timeit(lambda: ops._wna16_large(a, qw, qz, sc, None, 1))
                        row.append(f"mb{mb} wn{wn} ks{ks}: {t * 1e6:6.1f}{'' if err < 2e-3 else ' BAD %.1e' % err}")
                    except RuntimeError as ex:
                        row.append(f"mb{mb} wn{wn} ks{ks}: n/a")
        print(f"K={K:5d} N={N:5d} M={M:4d} | " + " | ".join(row), flush=True)
