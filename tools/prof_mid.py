"""Launches the mid-M W4A16 kernel (and the decode kernel beside it) on the four Llama-3-8B shapes for a rocprofv3
kernel trace:  rocprofv3 --kernel-trace -d gpurun_out/prof_mid -- python tools/prof_mid.py [M]; then tools/rocpd_stats.py."""
import os
import sys
import torch
from aphrodite_engine_amd import _custom_ops as ops
dev = "cuda"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 64
g = torch.Generator(device=dev).manual_seed(0)
for K, N in [(4096, 28672), (14336, 4096), (4096, 6144), (4096, 4096)]:
    qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // 8, N), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
    qz = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // 128, N // 8), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
    sc = (torch.rand(K // 128, N, generator=g, device=dev) * 0.01 + 0.005).half()
    a = torch.randn(M, K, device=dev, dtype=torch.float16, generator=g)
    for _ in range(10):
        ops._wna16_mid(a, qw, qz, sc, None, 1)
    os.environ["APHRO_WNA16_NO_LARGE"] = os.environ["APHRO_WNA16_NO_MID"] = "1"
    for _ in range(10):
        ops._wna16(a, qw, qz, sc, None, 1)
    del os.environ["APHRO_WNA16_NO_LARGE"], os.environ["APHRO_WNA16_NO_MID"]
    torch.cuda.synchronize()
