"""Times the mid-M W4A16 GEMM under each ablation library built by tools/mid_ablate.sh (one subprocess per library).
    python tools/mid_ablate.py [M] -- variants found under tools/bin/libmid_abl*.so"""
import glob
import os
import re
import subprocess
import sys

M = sys.argv[1] if len(sys.argv) > 1 else "64"
CHILD = r'''
import sys, torch
from aphrodite_engine_amd import _custom_ops as ops
M = int(sys.argv[1]); dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
out = []
for K, N in [(4096, 28672), (14336, 4096)]:
    qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // 8, N), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
    qz = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // 128, N // 8), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
    sc = (torch.rand(K // 128, N, generator=g, device=dev) * 0.01 + 0.005).half()
    a = torch.randn(M, K, device=dev, dtype=torch.float16, generator=g)
    fn = lambda: ops._wna16_mid(a, qw, qz, sc, None, 1)
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(20): fn()
    gr.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): gr.replay()
    e.record(); e.synchronize()
    out.append(f"K={K} N={N}: {s.elapsed_time(e) * 10:.1f} us")
print(" | ".join(out))
'''
libs = sorted(glob.glob("tools/bin/libmid_abl*.so"), key=lambda p: int(re.findall(r"abl(\d+)", p)[0]))
for lib in libs:
    env = dict(os.environ, APHRODITE_MI355X_LIB=os.path.abspath(lib), PYTHONPATH=".")
    r = subprocess.run([sys.executable, "-c", CHILD, M], env=env, capture_output=True, text=True)
    print(f"{os.path.basename(lib):22s} M={M}: {r.stdout.strip() or r.stderr.strip()[-300:]}", flush=True)
