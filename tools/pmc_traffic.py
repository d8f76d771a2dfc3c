#!/usr/bin/env python3
"""Reduce two rocprofv3 PMC passes over tools/prof_step_kernels.py (one with --pmc FETCH_SIZE, one with --pmc WRITE_SIZE;
never together: 3 + 2 TCC slots of 4, MI355X_MICROARCH.md "rocprofv3 PMC slots") into HBM bytes per launch of every
decode-step kernel, next to the algorithmic bytes of bench.py's roofline section.

    PMC_TAG=r5 python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> > profiles/r5_pmc_traffic.txt
    (also writes profiles/<PMC_TAG>_pmc_traffic.json; bench.py attaches the newest round's file as roofline.traffic with its provenance)

Units / corrections (guide, HBM section): FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports one half of
the bytes of wide coalesced streaming reads (128-byte requests tallied at 64): bytes = FETCH_SIZE * 1024 * 2.  WRITE_SIZE
is uncalibrated by the guide: it is calibrated HERE on a kernel with a known write volume (the fp32 slabs of the down
projection: 4 x 32 x 4096 x 4 B)."""
import csv
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(path, counter):
    acc = defaultdict(list)
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] == counter:
            acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def short(name):
    m = re.search(r"aphro::(\w+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    M, G = 32, 128

    def gemm_alg(K, N, m=M, out_bytes=0):
        return K * N // 2 + (K // G) * N * 2 + (K // G) * N // 2 + m * K * 2, out_bytes
    # (role in bench.py's roofline section, kernel-name regex, algorithmic read bytes, algorithmic write bytes)
    ctx, bs = 1100, 32
    kv = 2 * bs * ctx * 8 * 128 * 2
    table = [
        ("gate_up_proj", r"wna16_gemm_stream_kernel<2, 4, 8, 1, 3", *gemm_alg(4096, 28672, out_bytes=M * 14336 * 2)),
        ("gate_up_proj (two-pass resident kernel)", r"wna16_gemm_resident_kernel<2, 4, 8, 1, 3", *gemm_alg(4096, 28672, out_bytes=M * 14336 * 2)),
        ("gate_up_proj (round-2 kernel)", r"wna16_gemm_kernel<aphro::Half, 4, 2, 8>", *gemm_alg(4096, 28672, out_bytes=M * 14336 * 2)),
        ("down_proj", r"wna16_gemm_stream_kernel<2, 4, 7, 1, 0", *gemm_alg(14336, 4096, out_bytes=4 * M * 4096 * 4)),
        ("down_proj (round-2 kernel)", r"wna16_gemm_kernel<aphro::Half, 4, 2, 7>", *gemm_alg(14336, 4096, out_bytes=4 * M * 4096 * 4)),
        ("qkv_proj", r"wna16_gemm_stream_kernel<2, 4, 4, 1, 0", *gemm_alg(4096, 6144, out_bytes=2 * M * 6144 * 4)),
        ("qkv_proj (round-2 kernel)", r"wna16_gemm_kernel<aphro::Half, 4, 2, 4>", *gemm_alg(4096, 6144, out_bytes=2 * M * 6144 * 4)),
        ("o_proj", r"wna16_gemm_stream_kernel<2, 4, 2, 1, 0", *gemm_alg(4096, 4096, out_bytes=4 * M * 4096 * 4)),
        ("o_proj (round-2 kernel)", r"wna16_gemm_kernel<aphro::Half, 4, 2, 2>", *gemm_alg(4096, 4096, out_bytes=4 * M * 4096 * 4)),
        # round 6: the step's forms -- gate_up + SiluAndMul epilogue (16-bit activation [M, I] + partials out; mean over the plain
        # slab-form launches of the same template and the epilogue form), o / down quantising 16-bit pair-major A on load (<..., 1>)
        ("fp8_gate_up_proj", r"fp8_gemm_resident_kernel<aphro::Half, 2, 8, 7, 4, 0>", 4096 * 28672 + M * 4096, M * 14336 * 2 + M * 256 * 4),
        ("fp8_down_proj", r"fp8_gemm_resident_kernel<aphro::Half, 2, 7, 4, 6, 1>", 14336 * 4096 + M * 14336 * 2 + M * 256 * 4, 4 * M * 4096 * 4),
        ("fp8_down_proj (pre-quantised A, round 5)", r"fp8_gemm_resident_kernel<aphro::Half, 2, 7, 4, 8, 0>", 14336 * 4096 + M * 14336, 4 * M * 4096 * 4),
        ("fp8_qkv_proj", r"fp8_gemm_resident_kernel<aphro::Half, 2, 4, 3", 4096 * 6144 + M * 4096, 2 * M * 6144 * 4),
        ("fp8_o_proj", r"fp8_gemm_resident_kernel<aphro::Half, 2, 2, 4, 6, 1>", 4096 * 4096 + M * 4096 * 2 + M * 8 * 4, 4 * M * 4096 * 4),
        ("fp8_o_proj (pre-quantised A, round 5)", r"fp8_gemm_resident_kernel<aphro::Half, 2, 2, 4, 8, 0>", 4096 * 4096 + M * 4096, 4 * M * 4096 * 4),
        ("paged_attention", r"paged_attention_kernel<aphro::Half, 0, 128, 16, 8, 1", kv + 2 * bs * 6144 * 4, bs * 4096 * 2 + bs * 2 * 8 * 128 * 2),
        ("add_rms_norm_pack", r"add_rms_norm_pack_kernel", 4 * M * 4096 * 4 + M * 4096 * 2, 2 * M * 4096 * 2),
        ("gate_up_proj bs64 (mid kernel)", r"wna16_gemm_mid_kernel<2, 8, true>", *gemm_alg(4096, 28672, m=64, out_bytes=64 * 14336 * 2)),
        ("down_proj bs64 (mid kernel)", r"wna16_gemm_mid_kernel<2, 4, true>", *gemm_alg(14336, 4096, m=64, out_bytes=0)),
    ]

    def find(d, pat):
        for k, v in d.items():
            if re.search(pat.replace("<", "<").replace("(", r"\("), k):
                return v
        return None
    # calibrate WRITE_SIZE on the down projection's slabs
    wcal = None
    wd = find(write, r"wna16_gemm_kernel<aphro::Half, 4, 2, 7>")
    if wd:
        wcal = (4 * M * 4096 * 4) / (wd[0] * 1024)
    print("HBM traffic per launch of the decode-step kernels (round " + os.environ.get("PMC_TAG", "r4")[1:] + "; rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes of")
    print("tools/prof_step_kernels.py: bs 32, ctx 1100, configs[1] shapes; no trace domains in the same run)")
    print("bytes read = FETCH_SIZE [KiB] x 1024 x 2 (gfx950: wide coalesced reads are tallied at half their size -- MI355X_MICROARCH.md, HBM);")
    print(f"bytes written = WRITE_SIZE [KiB] x 1024 x {wcal:.3f} (calibrated on the down projection's 4 x 32 x 4096 fp32 slabs = 2 097 152 B)" if wcal else "WRITE_SIZE uncalibrated")
    print()
    print(f"{'role':34s} {'launches':>8s} {'read MB':>9s} {'alg read MB':>11s} {'ratio':>6s} {'write MB':>9s} {'alg write MB':>12s}")
    tag = os.environ.get("PMC_TAG", "r4")            # which round's files: profiles/<tag>_pmc_traffic.{txt,json}
    out = {"source": f"profiles/{tag}_pmc_traffic.txt: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of tools/prof_step_kernels.py "
                     f"(round {tag[1:]}, bs 32, ctx 1100); read bytes = FETCH_SIZE KiB x 1024 x 2 (gfx950 correction), write bytes calibrated on "
                     "the down projection's slabs", "kernels": {}}
    for role, pat, alg_r, alg_w in table:
        f, w = find(fetch, pat), find(write, pat)
        if f is None:
            print(f"{role:34s}  (no launch matched {pat})")
            continue
        rb = f[0] * 1024 * 2
        wb = w[0] * 1024 * wcal if (w and wcal) else float("nan")
        print(f"{role:34s} {f[1]:8d} {rb / 1e6:9.2f} {alg_r / 1e6:11.2f} {rb / alg_r:6.3f} {wb / 1e6:9.2f} {alg_w / 1e6:12.2f}")
        key = role.split(" ")[0]
        if " " not in role:
            out["kernels"][key] = {"hbm_bytes_per_launch": rb + (wb if wb == wb else 0.0), "read_bytes": rb, "write_bytes": wb if wb == wb else None,
                                   "algorithmic_read_bytes": alg_r, "algorithmic_write_bytes": alg_w}
    print()
    print("all aphro:: kernels seen (FETCH_SIZE KiB mean, launches):")
    for k, v in sorted(fetch.items(), key=lambda kv: -kv[1][0]):
        if "aphro::" in k:
            print(f"  {short(k):70s} {v[0]:12.1f} {v[1]:4d}")
    json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
