#!/usr/bin/env python3
"""What makes the same decode GEMM 8-10 % slower from one measurement to the next (profiles/r6_one_copy.txt (4), (8))?
The gate_up stream kernel at 32 rows over 16 weight sets (graph replay), measured
  (a) on four IDENTICAL sets of copies, all allocated up front and never released, in order and in reverse order  -> time / order
  (b) on a set allocated after 1 GB was released to the driver (the blocks of a dropped set)                      -> recycled memory
  (c) on copies placed +4 KiB into their blocks                                                                   -> base address
usage: python tools/base_offset_probe.py"""
import os
import sys
import time
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aphrodite_engine_amd import _custom_ops as ops  # noqa: E402

DEV = "cuda:0"
bs, K, N = 32, 4096, 28672


def make_sets(base, delta=0):
    sets = []
    for st, qz, sc in base:
        buf = torch.empty(st.numel() + (4 << 20) // 4, dtype=torch.int32, device=DEV)     # own block (> 10 MB: its own segment)
        v = buf[delta // 4: delta // 4 + st.numel()].view(st.shape)
        v.copy_(st)
        sets.append((v, qz, sc, buf))
    return sets


def capture(sets, packed):
    def run():
        for v, qz, sc, _ in sets:
            ops.wna16_gemm_resident(packed, bs, K, v, qz, sc, 1, mode="silu", strip_layout=True)
    run()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        run()
    gr.replay()
    torch.cuda.synchronize()
    return gr


def timed(gr, n_sets):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(5):
            gr.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / (5 * n_sets) * 1e3)
    return best


def main():
    g = torch.Generator(device=DEV).manual_seed(0)
    base = []
    for _ in range(16):
        qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // 8, N), generator=g, device=DEV, dtype=torch.int64).to(torch.int32)
        qz = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // 128, N // 8), generator=g, device=DEV, dtype=torch.int64).to(torch.int32)
        sc = (torch.rand(K // 128, N, generator=g, device=DEV) * 0.01 + 0.005).half()
        base.append((ops.wna16_strip_relayout(qw, bs, K // 128), qz, sc))
        del qw
    packed = ops.wna16_pack_a((torch.randn(bs, K, generator=g, device=DEV) * 0.5).half())
    print("gate_up stream kernel at 32 rows, us per launch (16 weight sets cycled, graph replay, best of 3 x 5)")
    four = [make_sets(base) for _ in range(4)]
    graphs = [capture(s, packed) for s in four]
    print("  (a) four identical sets, allocated up front:   in order " + "  ".join(f"{timed(gr, 16):.2f}" for gr in graphs)
          + "   reversed " + "  ".join(f"{timed(gr, 16):.2f}" for gr in reversed(graphs)))
    time.sleep(3.0)
    print("      after 3 s idle:                            in order " + "  ".join(f"{timed(gr, 16):.2f}" for gr in graphs))
    del graphs[3], four[3]
    torch.cuda.synchronize()
    torch.cuda.empty_cache()                           # ~1 GB back to the driver
    rec = make_sets(base)
    g_rec = capture(rec, packed)
    print(f"  (b) a set allocated after 1 GB was released: {timed(g_rec, 16):.2f}   (set 0 again: {timed(graphs[0], 16):.2f})")
    off = make_sets(base, 4096)
    g_off = capture(off, packed)
    print(f"  (c) copies +4 KiB into their blocks:          {timed(g_off, 16):.2f}   (set 0 again: {timed(graphs[0], 16):.2f})")
    print(f"      and the first measurement once more:      {timed(graphs[0], 16):.2f}  {timed(graphs[1], 16):.2f}  {timed(graphs[2], 16):.2f}")


if __name__ == "__main__":
    main()
