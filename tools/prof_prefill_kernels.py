"""Launches each prefill-side kernel a few times (for rocprofv3 --pmc FETCH_SIZE WRITE_SIZE, own pass)."""
import torch
from aphrodite_engine_amd import _custom_ops as ops
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
T, Hq, Hkv, D = 8192, 32, 8, 128
qkv = torch.randn(T, (Hq + 2 * Hkv) * D, device=dev, dtype=torch.float16, generator=g) * 0.5
q, k, v = qkv[:, :Hq * D].view(T, Hq, D), qkv[:, Hq * D:(Hq + Hkv) * D].view(T, Hkv, D), qkv[:, (Hq + Hkv) * D:].view(T, Hkv, D)
cu = torch.tensor([0, T], dtype=torch.int32, device=dev)
for _ in range(4):
    ops.flash_attn_varlen(q, k, v, cu, T, D ** -0.5, causal=True)
M, K, N = 8192, 4096, 28672
qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // 8, N), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
qz = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // 128, N // 8), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
sc = (torch.rand(K // 128, N, generator=g, device=dev) * 0.01 + 0.005).half()
a = torch.randn(M, K, device=dev, dtype=torch.float16, generator=g)
empty = torch.empty(0, dtype=torch.int32, device=dev)
for _ in range(4):
    ops.gptq_gemm(a, qw, qz, sc, empty, True, 4)
w8 = (torch.randn(N, K, device=dev, generator=g) * 0.5).to(torch.float8_e4m3fn)
a8 = a.to(torch.float8_e4m3fn)
sa = torch.rand(M, 1, device=dev, generator=g) * 0.1 + 0.05
sb = torch.rand(N, device=dev, generator=g) * 0.01 + 0.005
ob = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
for _ in range(4):
    ops.cutlass_scaled_mm(a8, w8.t(), sa, sb, torch.bfloat16, out=ob)
torch.cuda.synchronize()
