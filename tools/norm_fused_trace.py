#!/usr/bin/env python3
"""Per-wave wall-clock timeline of the norm-in-consumer gate_up / qkv launch (library built with
`make -C aphrodite_engine_amd/csrc ../lib/lab_norm_trace.so`, loaded through APHRODITE_MI355X_LIB): where the microseconds
of the hand-over go.  Stamps (100 MHz s_memrealtime, microseconds after the launch's first wave entry):
  0 entry | 1 producers: row stored and acknowledged | 2 staged weights landed and parked in LDS | 3 poll passed |
  4 k-step 0 issued (A fragments there) | 5 last LDS-fed k-step issued | 6 last k-step issued | 7 end
The launches run back to back over distinct weight copies inside ONE HIP graph; the last one stamps."""
import argparse
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aphrodite_engine_amd import _custom_ops as ops, _lib  # noqa: E402

NAMES = ["entry", "row acked (producers)", "staged W in LDS", "poll passed", "k-step 0 issued", "last LDS-fed k-step",
         "last k-step", "end"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", nargs="+", default=["gate_up"])
    ap.add_argument("--m", type=int, default=32)
    args = ap.parse_args()
    lib = _lib.lib()
    lib.aphro_wna16_resident_set_trace.argtypes = [ctypes.c_void_p]
    lib.aphro_wna16_resident_set_trace.restype = None
    M = args.m
    g = torch.Generator(device="cuda").manual_seed(0)
    for name in args.shapes:
        K, N = {"gate_up": (4096, 28672), "qkv": (4096, 6144)}[name]
        G = K // 128
        ncopy = max(3, (400 << 20) // (K * N // 2))
        qz = torch.randint(-2 ** 31, 2 ** 31 - 1, (G, N // 8), generator=g, device="cuda", dtype=torch.int64).to(torch.int32)
        sc = (torch.rand(G, N, generator=g, device="cuda") * 0.01).half()
        strips = [ops.wna16_strip_relayout(torch.randint(-2 ** 31, 2 ** 31 - 1, (K // 8, N), generator=g, device="cuda",
                                                         dtype=torch.int64).to(torch.int32), M, G) for _ in range(ncopy)]
        slabs = torch.randn(4, M, K, device="cuda") * 0.25
        res = torch.randn(M, K, device="cuda").half()
        w = torch.ones(K, device="cuda").half()
        sync = torch.zeros(ncopy, dtype=torch.int32, device="cuda")
        mode = "silu" if name == "gate_up" else "slabs"
        trace = torch.zeros(1024 * 4 * 8, dtype=torch.int64, device="cuda")

        def launches(stamp):
            sync.zero_()
            for i, st in enumerate(strips):
                lib.aphro_wna16_resident_set_trace(ctypes.c_void_p(trace.data_ptr() if (stamp and i == len(strips) - 1) else 0))
                ops.wna16_gemm_norm_fused(slabs, res, w, 1e-5, st, qz, sc, 1, sync[i:i + 1], mode=mode)
            lib.aphro_wna16_resident_set_trace(ctypes.c_void_p(0))
        launches(False)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            launches(True)
        for _ in range(3):
            gr.replay()
        torch.cuda.synchronize()
        s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_.record()
        for _ in range(5):
            gr.replay()
        e_.record()
        e_.synchronize()
        us = s_.elapsed_time(e_) * 1e3 / (5 * len(strips))
        t = trace.cpu().numpy().reshape(-1, 4, 8).astype(np.float64)
        nwg = int((t[:, 0, 0] != 0).sum())
        t = t[:nwg]
        t0 = t[:, :, 0].min()
        rel = (t - t0) / 100.0
        rel[t == 0] = np.nan
        rec = {"shape": name, "M": M, "workgroups": nwg, "graph_us_per_launch": round(us, 2), "timeline_us_p10_p50_p90_max": {}}
        prod = np.zeros(nwg, bool)
        prod[:M] = True
        for cls, mask in (("producers", prod), ("others", ~prod)):
            tl = {}
            for i, nm in enumerate(NAMES):
                x = rel[mask][:, :, i].ravel()
                x = x[~np.isnan(x)]
                if len(x):
                    tl[nm] = [round(float(np.percentile(x, q)), 2) for q in (10, 50, 90, 100)]
            rec["timeline_us_p10_p50_p90_max"][cls] = tl
        print(json.dumps(rec), flush=True)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "norm_fused_trace.jsonl"), "a") as f:
            f.write(json.dumps(rec) + "\n")


if __name__ == "__main__":
    main()
