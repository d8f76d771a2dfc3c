#!/usr/bin/env python3
"""Round-3: the resident-activation decode GEMM against the round-2 kernel at the configs[1] shapes, weights cold
(cycling over > 600 MB), HIP-graph replay, HIP-event timing.  --sweep adds the alternative plans, the strip-major layout
and (library built with -DRES_LAB) the prefetch-depth sweep."""
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
ALT = {"qkv": ["4,4,1,0", "4,4,0,3", "4,2,1,2"], "o": ["4,2,1,0", "4,1,2,0", "4,8,1,0"], "down": ["4,7,1,0", "7,2,2,0"],
       "gate_up": ["4,8,1,3"]}
OUT = []


def emit(**kw):
    OUT.append(kw)
    print(json.dumps(kw), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, nargs="+", default=[32])
    ap.add_argument("--sweep", action="store_true")
    args = ap.parse_args()
    g = torch.Generator(device="cuda").manual_seed(0)
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
        for M in args.m:
            a = torch.randn(M, K, device="cuda", dtype=torch.float16)
            pk = ops.wna16_pack_a(a)
            nb = wbytes + G * N * 2 + G * N // 2 + M * K * 2
            silu = name == "gate_up"

            def old():
                for qw, qz, sc in ws:
                    if silu:
                        ops.wna16_gemm_silu_pack(pk, M, K, qw, qz, sc, 1)
                    else:
                        ops.wna16_gemm_packed(pk, M, K, qw, qz, sc, 1, partials=True)
            tt = timeit(old, n)
            emit(kernel=name, M=M, impl="round2", us=round(tt * 1e6, 2), TBps=round(nb / tt / 1e12, 3))

            def res(wl, strip):
                for qw, qz, sc in wl:
                    ops.wna16_gemm_resident(pk, M, K, qw, qz, sc, 1, mode="silu" if silu else "slabs", strip_layout=strip)
            cfgs = [None] + (ALT[name] if args.sweep else [])
            for cfg in cfgs:
                if cfg is None:
                    os.environ.pop("APHRO_WNA16_RES_CFG", None)
                else:
                    os.environ["APHRO_WNA16_RES_CFG"] = cfg
                ks = ops.wna16_resident_ksplit(M, N, K, G)
                if ks <= 0 or (silu and ks != 1):
                    continue
                for strip in ((False, True) if args.sweep else (False, )):
                    wl = [(ops.wna16_strip_relayout(qw, M, G), qz, sc) for qw, qz, sc in ws] if strip else ws
                    depths = [None]
                    if args.sweep and M > 16 and cfg in (None, ) and name in ("gate_up", "down"):
                        depths += ["4,1", "8,2", "12,2"]
                    for d in depths:
                        if d is None:
                            os.environ.pop("APHRO_WNA16_RES_DEPTH", None)
                        else:
                            os.environ["APHRO_WNA16_RES_DEPTH"] = d
                        tt = timeit(lambda: res(wl, strip), n)
                        emit(kernel=name, M=M, impl="resident", cfg=cfg or "plan", ksplit=ks, strip=strip,
                             depth=d or "default", us=round(tt * 1e6, 2), TBps=round(nb / tt / 1e12, 3))
                    os.environ.pop("APHRO_WNA16_RES_DEPTH", None)
                    del wl
            os.environ.pop("APHRO_WNA16_RES_CFG", None)
        del ws
        torch.cuda.empty_cache()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "resident_bench.jsonl"), "w") as f:
        for r in OUT:
            f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
