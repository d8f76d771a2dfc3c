#!/usr/bin/env python3
"""Per-wave timeline of the resident decode GEMM (library built with `make RES_EXTRA=-DRES_LAB`): every wave stamps
s_memtime at entry / prologue issued / first k-step computed / end of pass 0 / end of the K loop / after the barrier / end,
plus the 100 MHz wall clock at entry and HW_ID / XCC_ID.  Prints the distribution of every phase over the waves."""
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

SHAPES = {"gate_up": (4096, 28672), "down": (14336, 4096), "qkv": (4096, 6144), "o": (4096, 4096)}
NAMES = ["entry", "prologue issued", "step 0 issued"] + [f"pass0 seg{s} MFMAs issued" for s in range(8)] + \
    ["last step issued", "K loop + flush done", "after barrier", "end"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", nargs="+", default=["gate_up", "down"])
    ap.add_argument("--variants", nargs="+", default=["plain:8,2", "strip:4,1", "strip:8,2", "strip:12,2"])
    ap.add_argument("--m", type=int, default=32)
    args = ap.parse_args()
    lib = _lib.lib()
    lib.aphro_wna16_resident_set_trace.argtypes = [ctypes.c_void_p]
    lib.aphro_wna16_resident_set_trace.restype = None
    M = args.m
    g = torch.Generator(device="cuda").manual_seed(0)
    out = []
    for name in args.shapes:
        K, N = SHAPES[name]
        G = K // 128
        ncopy = max(3, (400 << 20) // (K * N // 2))
        qws = [torch.randint(-2 ** 31, 2 ** 31 - 1, (K // 8, N), generator=g, device="cuda", dtype=torch.int64).to(torch.int32)
               for _ in range(ncopy)]
        qz = torch.randint(-2 ** 31, 2 ** 31 - 1, (G, N // 8), generator=g, device="cuda", dtype=torch.int64).to(torch.int32)
        sc = (torch.rand(G, N, generator=g, device="cuda") * 0.01).half()
        strips = [ops.wna16_strip_relayout(q, M, G) for q in qws]
        pk = ops.wna16_pack_a(torch.randn(M, K, device="cuda", dtype=torch.float16))
        mode = "silu" if name == "gate_up" else "slabs"
        for var in args.variants:
            layout, depth = var.split(":")
            os.environ["APHRO_WNA16_RES_DEPTH"] = depth       # (the TRACE instantiations exist for the swept depths only)
            ws = strips if layout == "strip" else qws
            trace = torch.zeros(1024 * 8 * 20, dtype=torch.int64, device="cuda")
            # back-to-back launches over distinct weight copies inside ONE HIP graph (code hot, weights cold: the decode
            # step's regime); only the LAST launch of the graph stamps (the trace pointer is baked in at capture)
            def launches():
                for i, w in enumerate(ws):
                    last = i == len(ws) - 1
                    lib.aphro_wna16_resident_set_trace(ctypes.c_void_p(trace.data_ptr() if last else 0))
                    ops.wna16_gemm_resident(pk, M, K, w, qz, sc, 1, mode=mode, strip_layout=layout == "strip")
                lib.aphro_wna16_resident_set_trace(ctypes.c_void_p(0))
            launches()
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                launches()
            s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            gr.replay()
            torch.cuda.synchronize()
            s_.record()
            for _ in range(5):
                gr.replay()
            e_.record()
            e_.synchronize()
            graph_us = s_.elapsed_time(e_) * 1e3 / (5 * len(ws))
            # and the same launches eagerly (host gaps between them)
            trace_e = torch.zeros_like(trace)
            for i, w in enumerate(ws):
                lib.aphro_wna16_resident_set_trace(ctypes.c_void_p(trace_e.data_ptr() if i == len(ws) - 1 else 0))
                ops.wna16_gemm_resident(pk, M, K, w, qz, sc, 1, mode=mode, strip_layout=layout == "strip")
            lib.aphro_wna16_resident_set_trace(ctypes.c_void_p(0))
            torch.cuda.synchronize()
            te = trace_e.cpu().numpy().reshape(-1, 20)
            te = te[te[:, 14] != 0]
            eager_span = float((te[:, 17].astype(np.float64).max() - te[:, 16].astype(np.float64).min()) / 100.0)
            t = trace.cpu().numpy().reshape(-1, 20)
            t = t[t[:, 14] != 0]
            st = t[:, :15].astype(np.float64)
            wall0, wall1 = t[:, 16].astype(np.float64), t[:, 17].astype(np.float64)
            rec = {"kernel": name, "variant": var, "waves": int(len(t)), "graph_us_per_launch": round(graph_us, 2),
                   "eager_wall_span_us": eager_span,
                   "wall_kernel_span_us": float((wall1.max() - wall0.min()) / 100.0),
                   "wall_entry_skew_us_max": float((wall0.max() - wall0.min()) / 100.0),
                   "wall_per_wave_us_p50_max": [float(np.percentile((wall1 - wall0) / 100.0, q)) for q in (50, 100)],
                   "GHz": float(np.median((st[:, 14] - st[:, 0]) / ((wall1 - wall0) * 10.0)))}
            # cumulative timeline: cycles since the wave's entry at each stamp (median / p90 / max over the waves)
            tl = {}
            for i, nm in enumerate(NAMES):
                if (st[:, i] == 0).all():
                    continue
                d = st[:, i] - st[:, 0]
                tl[nm] = [int(np.percentile(d, q)) for q in (50, 90, 100)]
            rec["cycles_since_entry_p50_p90_max"] = tl
            out.append(rec)
            print(json.dumps(rec), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "resident_trace.jsonl"), "w") as f:
        for r in out:
            f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
