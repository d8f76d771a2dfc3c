#!/usr/bin/env python3
"""Per-wave timeline of the decode attention kernel at the configs[1] geometry (library built with
`make PA_EXTRA=-DPA_LAB`): every wave stamps s_memtime at entry / first loads issued / data of pair i arrived / compute of
pair i issued / loop end / kernel end, plus the 100 MHz wall clock.  KV cold (three caches cycled inside one HIP graph:
> 256 MiB), the last launch of the graph stamps.  Forms: plain (q given) and fused (qkv slabs + rotary + cache write)."""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--ctx", type=int, default=1040)
    ap.add_argument("--kv", nargs="+", default=["auto", "fp8"])
    ap.add_argument("--forms", nargs="+", default=["plain", "fused"])
    args = ap.parse_args()
    lib = _lib.lib()
    lib.aphro_paged_attention_set_trace.argtypes = [ctypes.c_void_p]
    lib.aphro_paged_attention_set_trace.restype = None
    B, ctx, H, HKV, HD, BS = args.batch, args.ctx, 32, 8, 128, 16
    nblk_seq = (ctx + BS - 1) // BS
    nblocks = B * nblk_seq
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    out_rows = []
    for kv in args.kv:
        esz = 1 if kv != "auto" else 2
        x = 16 // esz
        caches = []
        for _ in range(3):
            if kv == "auto":
                kc = (torch.randn(nblocks, HKV, HD // x, BS, x, device=dev, generator=g) * 0.5).half()
                vc = (torch.randn(nblocks, HKV, HD, BS, device=dev, generator=g) * 0.5).half()
            else:
                kc = torch.randint(0, 120, (nblocks, HKV, HD // x, BS, x), device=dev, generator=g, dtype=torch.int32).to(torch.uint8)
                vc = torch.randint(0, 120, (nblocks, HKV, HD, BS), device=dev, generator=g, dtype=torch.int32).to(torch.uint8)
            caches.append((kc, vc))
        perm = torch.randperm(nblocks, device=dev, generator=g).to(torch.int32)
        bt = perm.view(B, nblk_seq).contiguous()
        seq_lens = torch.full((B, ), ctx, dtype=torch.int32, device=dev)
        q = (torch.randn(B, H, HD, device=dev, generator=g) * 0.5).half()
        ntot = (H + 2 * HKV) * HD
        slabs = torch.randn(2, B, ntot, device=dev, generator=g) * 0.3
        cos_sin = torch.randn(B, HD, device=dev, generator=g).half()
        slot = (bt[:, -1].long() * BS + (ctx - 1) % BS)
        scale = HD ** -0.5
        for form in args.forms:
            trace = torch.zeros(HKV * B * 8 * 20, dtype=torch.int64, device=dev)

            def launches(tr):
                for i, (kc, vc) in enumerate(caches):
                    lib.aphro_paged_attention_set_trace(ctypes.c_void_p(tr.data_ptr() if i == len(caches) - 1 else 0))
                    if form == "plain":
                        o = torch.empty_like(q)
                        ops.paged_attention_v1(o, q, kc, vc, HKV, scale, bt, seq_lens, BS, ctx, None, kv, 1.0, 1.0)
                    else:
                        ops.paged_attention_rope_packed(slabs, None, cos_sin, slot, kc, vc, H, HKV, scale, bt, seq_lens, BS, ctx,
                                                        None, kv, 1.0, 1.0)
                lib.aphro_paged_attention_set_trace(ctypes.c_void_p(0))
            launches(trace)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                launches(trace)
            gr.replay()
            torch.cuda.synchronize()
            s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s_.record()
            for _ in range(5):
                gr.replay()
            e_.record()
            e_.synchronize()
            us = s_.elapsed_time(e_) * 1e3 / (5 * len(caches))
            t = trace.cpu().numpy().reshape(-1, 20).astype(np.float64)
            t = t[t[:, 0] > 0]
            if len(t) == 0:
                print(json.dumps({"kv": kv, "form": form, "us_per_launch": round(us, 2), "error": "no stamps: library not built with PA_LAB"}))
                continue
            wall0 = t[:, 16].min()
            start = (t[:, 16] - wall0) / 100.0          # us since the first wave of the launch started
            end = (t[:, 17] - wall0) / 100.0
            # cycle stamps -> us with the wall-clock span of the same wave as the ruler
            cyc = t[:, 15] - t[:, 0]
            scale_us = (t[:, 17] - t[:, 16]) / 100.0 / np.maximum(cyc, 1)
            rel = (t[:, :16] - t[:, :1]) * scale_us[:, None]
            nit = t[:, 19]
            names = ["entry", "first loads issued"]
            for i in range(6):
                names += [f"pair {i} data arrived", f"pair {i} compute issued"]
            names = names[:11] + ["fused: q slab loads issued", "fused: first K/V loads issued", "fused: own q share in LDS", "loop done", "end"]
            names[1] = "plain: first loads issued / fused: q-phase barrier passed"
            row = {"kv": kv, "form": form, "us_per_launch_in_graph": round(us, 2), "waves": int(len(t)),
                   "wave start us (p50/p90/max)": [round(float(np.percentile(start, q_)), 2) for q_ in (50, 90, 100)],
                   "wave end us (p10/p50/max)": [round(float(np.percentile(end, q_)), 2) for q_ in (10, 50, 100)],
                   "pairs per wave (min/max)": [int(nit.min()), int(nit.max())]}
            print(json.dumps(row))
            for i, nm in enumerate(names):
                col = rel[:, i]
                ok = t[:, i] > 0
                if ok.sum() == 0:
                    continue
                print(f"    {nm:26s} p10 {np.percentile(col[ok], 10):6.2f}  p50 {np.percentile(col[ok], 50):6.2f}  p90 {np.percentile(col[ok], 90):6.2f} us"
                      f"   ({int(ok.sum())} waves)")
            out_rows.append(row)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "attn_trace.json"), "w") as f:
        json.dump(out_rows, f, indent=1)


if __name__ == "__main__":
    main()
