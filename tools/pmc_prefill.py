#!/usr/bin/env python3
"""Reduce the three rocprofv3 PMC passes over tools/prof_prefill_r6.py (FETCH_SIZE; WRITE_SIZE; the SQ set) into
profiles/<tag>_pmc_prefill.{txt,json}: HBM-side bytes and MFMA-pipe utilisation of the prefill kernels, per launch.

    python tools/pmc_prefill.py <dir with pmc_prefill_FETCH_SIZE/, pmc_prefill_WRITE_SIZE/, pmc_prefill_SQ_VALU_MFMA_BUSY_CYCLES/> r6

Units (MI355X_MICROARCH.md, HBM + "rocprofv3 PMC slots"): FETCH_SIZE / WRITE_SIZE in KiB, FETCH_SIZE doubled on gfx950 (wide
coalesced reads are tallied at half their size); SQ_VALU_MFMA_BUSY_CYCLES in shader cycles summed over every SIMD that ran an MFMA
(16 x 16 x 32 f16: 8 passes = 32 cycles on gfx950? -- NOT assumed: the utilisation below is busy cycles / (SIMDs x kernel cycles),
kernel cycles from GRBM_GUI_ACTIVE of the same pass (summed over the 8 XCDs by rocprofv3: / 8); SQ_WAVE_CYCLES / SQ_WAIT_* /
SQ_ACTIVE_INST_* count quad-cycles summed over waves."""
import csv
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(4096, 6144, "qkv"), (4096, 4096, "o"), (4096, 28672, "gate_up"), (14336, 4096, "down")]
M, T, HQ, HKV, D = 8192, 8192, 32, 8, 128
NL = int(os.environ.get("PROF_NL", "3"))


def load(path):
    """{kernel short name: [ {counter: value, 'dur_ns': ...} per dispatch, in dispatch order ]}"""
    per = defaultdict(dict)
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        m = re.search(r"aphro::(\w+)", name)
        if not m:
            continue
        d = per[(int(r["Dispatch_Id"]), m.group(1))]
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        d["dur_ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    out = defaultdict(list)
    for (did, k), d in sorted(per.items()):
        out[k].append(d)
    return out


def groups(launches):
    """launches of one kernel in program order -> list of per-shape lists without the first (cold code) launch of each"""
    n = NL + 1
    return [launches[i * n + 1:(i + 1) * n] for i in range(len(launches) // n)]


def mean(xs):
    return sum(xs) / max(1, len(xs))


def main():
    base, tag = sys.argv[1], sys.argv[2]
    f = load(os.path.join(base, "pmc_prefill_FETCH_SIZE", "p_counter_collection.csv"))
    w = load(os.path.join(base, "pmc_prefill_WRITE_SIZE", "p_counter_collection.csv"))
    q = load(os.path.join(base, "pmc_prefill_SQ_VALU_MFMA_BUSY_CYCLES", "p_counter_collection.csv"))
    rows = []

    def add(label, kernel, gi, alg_read, alg_write, flops, peak_tf):
        fg, wg, qg = groups(f[kernel])[gi], groups(w[kernel])[gi], groups(q[kernel])[gi]
        read = mean([d["FETCH_SIZE"] for d in fg]) * 1024 * 2
        write = mean([d["WRITE_SIZE"] for d in wg]) * 1024
        busy = mean([d["SQ_VALU_MFMA_BUSY_CYCLES"] for d in qg])
        gui = mean([d["GRBM_GUI_ACTIVE"] for d in qg]) / 8.0
        dur = mean([d["dur_ns"] for d in qg])
        wc = mean([d["SQ_WAVE_CYCLES"] for d in qg])
        rows.append({
            "kernel": kernel, "case": label, "launches": len(fg),
            "read_bytes": read, "write_bytes": write, "alg_read_bytes": alg_read, "alg_write_bytes": alg_write,
            "read_ratio": read / alg_read, "write_ratio": write / alg_write,
            "mfma_busy_cycles": busy, "gui_active_cycles_per_xcd": gui, "mfma_util": busy / (gui * 1024.0),
            "dur_us_under_pmc": dur / 1e3, "clock_ghz_under_pmc": gui / dur,
            "tflops_under_pmc": flops / dur / 1e3, "frac_of_peak_under_pmc": flops / dur / 1e3 / peak_tf,
            "sq_busy_cycles": mean([d["SQ_BUSY_CYCLES"] for d in qg]),
            "wave_cycles": wc,
            "wait_any_frac": mean([d["SQ_WAIT_ANY"] for d in qg]) / wc,
            "wait_inst_any_frac": mean([d["SQ_WAIT_INST_ANY"] for d in qg]) / wc,
            "active_inst_any_frac": mean([d["SQ_ACTIVE_INST_ANY"] for d in qg]) / wc,
            "active_inst_valu_frac": mean([d["SQ_ACTIVE_INST_VALU"] for d in qg]) / wc,
        })

    qb, kvb = T * HQ * D * 2, 2 * T * HKV * D * 2
    add(f"T={T} causal Hq {HQ} / Hkv {HKV}", "flash_attn_varlen_v4_kernel", 0, qb + kvb, qb, 4.0 * T * T * D * HQ / 2, 2500.0)
    for gi, (K, N, nm) in enumerate(SHAPES):
        wbytes = K * N // 2 + (K // 128) * N * 2 + (K // 128) * N // 2
        add(f"W4A16 {nm} {M}x{K}x{N}", "wna16_gemm_large8_kernel", gi, M * K * 2 + wbytes, M * N * 2, 2.0 * M * K * N, 2500.0)
        add(f"W8A8 {nm} {M}x{K}x{N}", "fp8_gemm_large8_kernel", gi, M * K + K * N + (M + N) * 4, M * N * 2, 2.0 * M * K * N, 5000.0)
    lines = [
        f"HBM-side traffic and MFMA-pipe utilisation of the prefill kernels (round {tag[1:]}): rocprofv3 --pmc, THREE separate passes of",
        "tools/prof_prefill_r6.py (FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY",
        "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE), --kernel-trace only beside them; a 512 MB fill between launches evicts the",
        f"Infinity Cache; means over {NL} launches after one discarded.  read = FETCH_SIZE KiB x 1024 x 2 (gfx950 correction of the guide),",
        "write = WRITE_SIZE KiB x 1024.  mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs): the share of",
        "SIMD-cycles in which the matrix pipe was busy.  wait / active columns are shares of SQ_WAVE_CYCLES (quad-cycles, disjoint:",
        "WAIT_ANY = parked on s_waitcnt / barrier, WAIT_INST_ANY = issue stall, ACTIVE_INST_ANY = issuing).  Durations are those of the",
        "counter pass (serialised dispatches, counters armed) -- the timed numbers are bench.py's prefill_kernels.",
        "",
        f"{'case':38s} {'read MB':>9s} {'alg':>8s} {'ratio':>6s} {'write MB':>9s} {'alg':>8s} {'mfma_util':>9s} {'us(pmc)':>8s} {'GHz':>5s} {'TF/s':>6s} {'wait':>5s} {'stall':>5s} {'issue':>5s} {'valu':>5s}",
    ]
    for r in rows:
        lines.append(f"{r['case']:38s} {r['read_bytes'] / 1e6:9.1f} {r['alg_read_bytes'] / 1e6:8.1f} {r['read_ratio']:6.2f} "
                     f"{r['write_bytes'] / 1e6:9.1f} {r['alg_write_bytes'] / 1e6:8.1f} {r['mfma_util']:9.3f} {r['dur_us_under_pmc']:8.1f} "
                     f"{r['clock_ghz_under_pmc']:5.2f} {r['tflops_under_pmc']:6.0f} {r['wait_any_frac']:5.2f} {r['wait_inst_any_frac']:5.2f} "
                     f"{r['active_inst_any_frac']:5.2f} {r['active_inst_valu_frac']:5.2f}")
    txt = "\n".join(lines) + "\n"
    sys.stdout.write(txt)
    open(os.path.join(ROOT, "profiles", f"{tag}_pmc_prefill.txt"), "w").write(txt)
    json.dump({"source": f"profiles/{tag}_pmc_prefill.txt (tools/pmc_prefill.py over three rocprofv3 --pmc passes of tools/prof_prefill_r6.py)",
               "rows": rows}, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_prefill.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
