"""Lab: s_memtime stamps of the stream-K fp8 GEMM's segments (APHRO_FP8_LARGE_DEBUG=4)."""
import os, sys
os.environ["APHRO_FP8_LARGE_DEBUG"] = "4"
os.environ["APHRO_FP8_LARGE_STREAMK"] = "1"
import torch
from aphrodite_engine_amd import _custom_ops as ops
DEV = "cuda"
M, N, K = [int(x) for x in sys.argv[1:4]] if len(sys.argv) > 3 else (8192, 28672, 4096)
g = torch.Generator(device=DEV).manual_seed(0)
w = (torch.randn(N, K, generator=g, device=DEV) * 0.5).to(torch.float8_e4m3fn)
a = torch.randn(M, K, generator=g, device=DEV).to(torch.float8_e4m3fn)
sb = torch.rand(N, generator=g, device=DEV) * 0.01 + 0.005
sa = torch.rand(M, 1, generator=g, device=DEV) * 0.1 + 0.05
out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
for _ in range(3):
    ops.cutlass_scaled_mm(a, w.t(), sa, sb, torch.bfloat16, out=out)
torch.cuda.synchronize()
ws = ops._workspace(a.device, ops._lib.lib().aphro_scaled_mm_fp8_large_workspace_bytes(M, N, K))
st = ws[2048:2048 + 8 * 16 * 8].view(torch.int64).cpu().view(8, 2, 8)
names = ["seg start", "1st tile landed", "K loop done", "epi_put done", "stores issued", "stores acked"]
for wg in range(8):
    for sg in range(2):
        t = st[wg, sg]
        print(f"wg {wg} seg {sg + 1}: " + "  ".join(f"{names[i]} +{int(t[i] - t[0]):6d}" for i in range(1, 6)))
