#!/usr/bin/env python3
"""HBM-side bytes of the prompt-sized W4A16 GEMM on [K/8, N] and on the strip-major copy read in place (round 6), for
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -- python tools/prof_strip_fetch.py
M = 4096 (eight-phase fused form) on the four Llama-3-8B shapes and M = 1024 on qkv / down (the round-2 tile machine's K-sliced
plans); a 512 MB fill between launches evicts the Infinity Cache.  `python tools/prof_strip_fetch.py --reduce <dir>` prints the table."""
import csv
import glob
import os
import sys
from collections import defaultdict

SHAPES = ((4096, 6144), (4096, 4096), (4096, 28672), (14336, 4096))


def reduce(d):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == "FETCH_SIZE" and "wna16_gemm_large" in r["Kernel_Name"]]
    per = defaultdict(float)
    meta = {}
    for r in rows:
        per[int(r["Dispatch_Id"])] += float(r["Counter_Value"])
        meta[int(r["Dispatch_Id"])] = r["Kernel_Name"].split("(")[0].replace("void aphro::", "")
    ids = sorted(per)
    # launch order of main(): for every (M, K, N): 1 warm + 3 x [K/8, N], then 1 warm + 3 x strip-major
    print("FETCH_SIZE per launch (KiB x 1024 x 2 = bytes, the guide's gfx950 correction), mean of 3 launches after one warm-up")
    print(f"{'M':>6} {'K x N':>14} {'kernel':>52} {'[K/8,N] MB':>12} {'strip MB':>10} {'ratio':>7} {'algorithmic MB':>15}")
    i = 0
    for M in (4096, 1024):
        for K, N in (SHAPES if M == 4096 else (SHAPES[0], SHAPES[3])):
            grp = ids[i:i + 8]
            i += 8
            rm = sum(per[j] for j in grp[1:4]) / 3 * 1024 * 2 / 1e6
            st = sum(per[j] for j in grp[5:8]) / 3 * 1024 * 2 / 1e6
            alg = (M * K * 2 + K * N / 2 + (K // 128) * N * 2.5) / 1e6
            print(f"{M:>6} {str(K) + ' x ' + str(N):>14} {meta[grp[5]][:52]:>52} {rm:>12.1f} {st:>10.1f} {st / rm:>7.3f} {alg:>15.1f}")


def main():
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from aphrodite_engine_amd import _custom_ops as ops
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    for M in (4096, 1024):
        for K, N in (SHAPES if M == 4096 else (SHAPES[0], SHAPES[3])):
            qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // 8, N), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
            qz = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // 128, N // 8), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
            sc = (torch.rand(K // 128, N, generator=g, device=dev) * 0.01 + 0.005).half()
            st = ops.wna16_strip_relayout(qw, 32, K // 128)
            a = torch.randn(M, K, device=dev, dtype=torch.float16, generator=g)
            with ops.knob("APHRO_WNA16_LARGE_TWO_PASS", 0):
                for _ in range(4):
                    flush.fill_(1)
                    ops._wna16_large(a, qw, qz, sc, None, 1)
                for _ in range(4):
                    flush.fill_(1)
                    ops.wna16_gemm_large_strip(a, st, qz, sc, 1)
            del qw, qz, sc, st, a
    torch.cuda.synchronize()


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--reduce":
        reduce(sys.argv[2])
    else:
        main()
