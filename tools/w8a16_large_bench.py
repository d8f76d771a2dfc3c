"""W8A16 prefill-sized GEMM (fp8 weights, f16 activations) vs widen + library GEMM.  python tools/w8a16_large_bench.py"""
import os, torch
from aphrodite_engine_amd import _custom_ops as ops
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) * 1e-3 / iters
for K, N in [(4096, 6144), (4096, 28672), (14336, 4096)]:
    w = (torch.randn(N, K, generator=g, device=dev) * 0.5).to(torch.float8_e4m3fn)
    sb = torch.rand(N, generator=g, device=dev) * 0.01 + 0.005
    for M in (256, 2048, 8192):
        a = torch.randn(M, K, generator=g, device=dev).half()
        t_new = timeit(lambda: ops.fp8_marlin_gemm(a, w, sb, None, 8, M, N, K))
        os.environ["APHRO_WNA16_NO_LARGE"] = "1"
        t_lib = timeit(lambda: ops.fp8_marlin_gemm(a, w, sb, None, 8, M, N, K)) if M >= 256 else float("nan")
        del os.environ["APHRO_WNA16_NO_LARGE"]
        fl = 2.0 * M * N * K
        print(f"K={K:5d} N={N:5d} M={M:5d}: large kernel {t_new*1e6:9.1f} us {fl/t_new/1e12:7.1f} TF | widen + library GEMM {t_lib*1e6:9.1f} us {fl/t_lib/1e12:7.1f} TF", flush=True)
