#!/usr/bin/env python3
"""Round-3: the OP-LEVEL decode GEMM (`ops.gptq_gemm`, row-major f16 in, [M, N] out) at the configs[1] shapes: the
one-launch form (row-major A gathered in the kernel + in-kernel K-slice reduce, csrc/wna16_gemm_resident.hip) against the
pack + GEMM (+ reduce) launches it replaces (APHRO_WNA16_OP_NO_RESIDENT=1).  Weights cold (cycling over > 600 MB), HIP-graph
replay, HIP-event timing; per call = everything the op launches."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aphrodite_engine_amd import _custom_ops as ops  # noqa: E402
from tools.prefetch_lab import timeit  # noqa: E402

SHAPES = {"gate_up": (4096, 28672), "down": (14336, 4096), "qkv": (4096, 6144), "o": (4096, 4096)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, nargs="+", default=[32, 8, 1])
    args = ap.parse_args()
    g = torch.Generator(device="cuda").manual_seed(0)
    g_idx = torch.empty(0, dtype=torch.int32, device="cuda")
    out = []
    for name, (K, N) in SHAPES.items():
        G = K // 128
        wbytes = K * N // 2
        n = max(2, (640 << 20) // wbytes)
        ws = []
        for _ in range(n):
            qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // 8, N), generator=g, device="cuda", dtype=torch.int64).to(torch.int32)
            qz = torch.randint(-2 ** 31, 2 ** 31 - 1, (G, N // 8), generator=g, device="cuda", dtype=torch.int64).to(torch.int32)
            sc = (torch.rand(G, N, generator=g, device="cuda") * 0.01).half()
            ws.append((qw, qz, sc))
        strips = None
        for M in args.m:
            a = torch.randn(M, K, device="cuda", dtype=torch.float16)
            nb = wbytes + G * N * 2 + G * N // 2 + M * K * 2 + M * N * 2

            def op():
                for qw, qz, sc in ws:
                    ops.gptq_gemm(a, qw, qz, sc, g_idx, True, 4)
            for impl in ("three-launch", "one-launch", "one-launch strip-major"):
                if impl == "three-launch":
                    os.environ["APHRO_WNA16_OP_NO_RESIDENT"] = "1"
                else:
                    os.environ.pop("APHRO_WNA16_OP_NO_RESIDENT", None)
                    if not ops.wna16_gemm_rowmajor_supported(M, N, K, G, torch.float16):
                        continue
                if impl.endswith("strip-major"):
                    if strips is None:
                        strips = [(ops.wna16_strip_relayout(qw, M, G), qz, sc) for qw, qz, sc in ws]

                    def fn():
                        for qw, qz, sc in strips:
                            ops.wna16_gemm_rowmajor(a, qw, qz, sc, 1, strip_layout=True)
                else:
                    fn = op
                fn()
                tt = timeit(fn, n)
                r = dict(kernel=name, M=M, impl=impl, us=round(tt * 1e6, 2), TBps=round(nb / tt / 1e12, 3))
                out.append(r)
                print(json.dumps(r), flush=True)
            os.environ.pop("APHRO_WNA16_OP_NO_RESIDENT", None)
        del ws, strips
        torch.cuda.empty_cache()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "op_gemm_bench.jsonl"), "w") as f:
        for r in out:
            f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
