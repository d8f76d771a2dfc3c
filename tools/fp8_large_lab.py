"""Lab: raw kernel time of aphro_scaled_mm_fp8_large (HIP-graph of 10 launches, no Python in the timed region).
   python tools/fp8_large_lab.py [M N K ...]"""
import sys
import torch
from aphrodite_engine_amd import _custom_ops as ops

DEV = "cuda"
shapes = [(8192, 4096, 4096), (8192, 6144, 4096), (8192, 28672, 4096), (8192, 4096, 14336), (2048, 6144, 4096), (1024, 6144, 4096)]
if len(sys.argv) > 3:
    v = [int(x) for x in sys.argv[1:]]
    shapes = [tuple(v[i:i + 3]) for i in range(0, len(v), 3)]
g = torch.Generator(device=DEV).manual_seed(0)
for M, N, K in shapes:
    w = (torch.randn(N, K, generator=g, device=DEV) * 0.5).to(torch.float8_e4m3fn)
    a = torch.randn(M, K, generator=g, device=DEV).to(torch.float8_e4m3fn)
    sb = torch.rand(N, generator=g, device=DEV) * 0.01 + 0.005
    sa = torch.rand(M, 1, generator=g, device=DEV) * 0.1 + 0.05
    out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    fn = lambda: ops.cutlass_scaled_mm(a, w.t(), sa, sb, torch.bfloat16, out=out)
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(10):
            fn()
    gr.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); gr.replay(); gr.replay(); e.record(); e.synchronize()
    t = s.elapsed_time(e) * 1e-3 / 20
    print(f"M={M} N={N} K={K}: {t*1e6:8.1f} us  {2.0*M*N*K/t/1e12:7.1f} TF", flush=True)
