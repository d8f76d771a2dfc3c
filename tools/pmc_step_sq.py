#!/usr/bin/env python3
"""Reduce two rocprofv3 PMC passes over tools/prof_step_kernels.py -- the SQ set (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES
SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE) and the TCP set
(TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum) -- into where a decode kernel's
wave cycles go ("issue-bound" or "data-bound").

    python tools/pmc_step_sq.py <dir with pmc_step_SQ/ and pmc_step_TCP/> > profiles/r6_pmc_step_sq.txt

SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves and are disjoint (MI355X_MICROARCH.md,
"rocprofv3 PMC slots"): WAIT_ANY = parked on s_waitcnt / a barrier (data), WAIT_INST_ANY = issue stall (MFMA read-after-write,
a busy pipe), ACTIVE_INST_ANY = issuing (of which ACTIVE_INST_VALU)."""
import collections
import csv
import re
import sys


def load(path):
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(path)):
        m = re.search(r"aphro::(\w+)(<[^>]*>)?", r["Kernel_Name"])
        if not m:
            continue
        d = per[(int(r["Dispatch_Id"]), m.group(1) + (m.group(2) or ""))]
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        d["dur"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    out = collections.defaultdict(list)
    for (_, k), d in sorted(per.items()):
        out[k].append(d)
    return out


def mean(xs):
    return sum(xs) / len(xs)


ROLE = {"wna16_gemm_stream_kernel<2, 4, 8, 1, 3, 4>": "int4 gate_up + SiluAndMul (step kernel)",
        "wna16_gemm_stream_kernel<2, 4, 7, 1, 0, 6>": "int4 down (step kernel)",
        "wna16_gemm_stream_kernel<2, 4, 4, 1, 0, 6>": "int4 qkv (step kernel)",
        "wna16_gemm_stream_kernel<2, 4, 2, 1, 0, 4>": "int4 o (step kernel)",
        "paged_attention_kernel<aphro::Half, 0, 128, 16, 8, 1, false>": "decode attention ctx 1100 (fused rope form)",
        "add_rms_norm_pack_kernel<aphro::Half, false, false, 4>": "norm + pack",
        "lm_head_argmax_kernel<aphro::Half, 2, 4>": "LM head + argmax",
        "fp8_gemm_resident_kernel<aphro::Half, 2, 8, 7, 4>": "fp8 gate_up", "fp8_gemm_resident_kernel<aphro::Half, 2, 7, 4, 8>": "fp8 down",
        "fp8_gemm_resident_kernel<aphro::Half, 2, 4, 3, 8>": "fp8 qkv", "fp8_gemm_resident_kernel<aphro::Half, 2, 2, 4, 8>": "fp8 o"}


def main():
    base = sys.argv[1]
    q = load(base + "/pmc_step_SQ/p_counter_collection.csv")
    t = load(base + "/pmc_step_TCP/p_counter_collection.csv")
    print(__doc__.split("\n\n")[0].replace("Reduce two", "Two") + "\n")
    print("shares of SQ_WAVE_CYCLES per kernel (means over the launches after the first); mfma = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x")
    print("GRBM_GUI_ACTIVE / 8); tcp_stall = TCP_PENDING_STALL_CYCLES / TCP_GATE_EN1_sum (vector-L1 cycles stalled on its pending-request queue)\n")
    print(f"{'kernel':64s} {'role':44s} {'wait':>5s} {'stall':>5s} {'issue':>5s} {'valu':>5s} {'mfma':>6s} {'tcp_stall':>9s}")
    for k, ls in q.items():
        if k not in ROLE:
            continue
        ls = ls[1:] if len(ls) > 1 else ls
        wc = mean([d["SQ_WAVE_CYCLES"] for d in ls])
        gui = mean([d["GRBM_GUI_ACTIVE"] for d in ls]) / 8
        tl = t.get(k, [])[1:]
        ts = mean([d["TCP_PENDING_STALL_CYCLES"] for d in tl]) / mean([d["TCP_GATE_EN1_sum"] for d in tl]) if tl else float("nan")
        print(f"{k:64s} {ROLE[k]:44s} {mean([d['SQ_WAIT_ANY'] for d in ls]) / wc:5.2f} {mean([d['SQ_WAIT_INST_ANY'] for d in ls]) / wc:5.2f} "
              f"{mean([d['SQ_ACTIVE_INST_ANY'] for d in ls]) / wc:5.2f} {mean([d['SQ_ACTIVE_INST_VALU'] for d in ls]) / wc:5.2f} "
              f"{mean([d['SQ_VALU_MFMA_BUSY_CYCLES'] for d in ls]) / (gui * 1024):6.3f} {ts:9.2f}")


if __name__ == "__main__":
    main()
