#!/usr/bin/env python3
"""Round-3 lab: the grouped (MoE) int4 GEMM launch streams its weights at ~7 TB/s in the Mixtral step
(profiles: wna16_gemm_kernel<Half, 4, 1, 8> 63.8 us for 470 MB) while the dense decode GEMM of the same template family
reaches ~4.5 TB/s in its K loop.  Is it the many small 16-row workgroups per CU?  Run the DENSE gate_up / down GEMM (M = 32)
through the grouped entry as two 16-row blocks of one "expert" (same weights) and through E = 1..8 distinct experts of 16
rows each, weights cold, graph replay."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aphrodite_engine_amd import _custom_ops as ops  # noqa: E402
from tools.prefetch_lab import timeit  # noqa: E402


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    for name, (K, N) in {"gate_up": (4096, 28672), "down": (14336, 4096)}.items():
        G = K // 128
        wbytes = K * N // 2
        ncopy = max(2, (640 << 20) // wbytes)
        E = 8
        qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (ncopy, E, K // 8, N), generator=g, device="cuda", dtype=torch.int64).to(torch.int32) \
            if wbytes * ncopy * E < (20 << 30) else None
        qz = torch.randint(-2 ** 31, 2 ** 31 - 1, (E, G, N // 8), generator=g, device="cuda", dtype=torch.int64).to(torch.int32)
        sc = (torch.rand(E, G, N, generator=g, device="cuda") * 0.01).half()
        a = torch.randn(32, K, device="cuda", dtype=torch.float16)
        pk = ops.wna16_pack_a(a)
        # dense reference
        def dense():
            for c in range(ncopy):
                ops.wna16_gemm_packed(pk, 32, K, qw[c, 0], qz[0], sc[0], 1, partials=True)
        tt = timeit(dense, ncopy)
        print(json.dumps(dict(kernel=name, impl="dense round-2 kernel, M=32", us=round(tt * 1e6, 2), TBps=round(wbytes / tt / 1e12, 3))), flush=True)
        for nexp, rows_per in ((1, 32), (2, 16), (4, 16), (8, 16)):
            # m-blocks of 16 rows: nexp experts; for nexp == 1: two blocks on the same expert (the dense GEMM as a grouped launch)
            nblk = 2 if nexp == 1 else nexp
            m_pad = 16 * nblk
            eid = torch.tensor([0, 0] if nexp == 1 else list(range(nexp)), dtype=torch.int32, device="cuda")
            post = torch.tensor([m_pad], dtype=torch.int32, device="cuda")
            ap = ops.wna16_pack_a(torch.randn(m_pad, K, device="cuda", dtype=torch.float16))
            def grouped():
                for c in range(ncopy):
                    ops.wna16_gemm_grouped(ap, m_pad, K, qw[c], qz, sc, eid, post, 1, "slabs")
            tt = timeit(grouped, ncopy)
            nb = wbytes * nexp
            print(json.dumps(dict(kernel=name, impl=f"grouped: {nblk} blocks of 16 rows over {nexp} expert(s)", us=round(tt * 1e6, 2),
                                  TBps=round(nb / tt / 1e12, 3))), flush=True)
        del qw
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
