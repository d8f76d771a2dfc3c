#!/usr/bin/env python3
"""Reduce `rocprofv3 --pmc ... --kernel-trace --output-format csv` runs of tools/prof_placement.py: per decode-GEMM grid
(= projection), mean duration and mean counter values before (A) and after (B) the last relayout dispatch.
usage: prof_placement_reduce.py <dir> [<dir> ...]"""
import csv
import glob
import sys
from collections import defaultdict

GRIDS = {"256": "gate_up", "128": "down/o", "96": "qkv"}


def load(d):
    trace = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    pmc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    rows = {}
    if trace:
        for r in csv.DictReader(open(trace[0])):
            rows[int(r["Dispatch_Id"])] = dict(name=r["Kernel_Name"], dur=int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
                                               grid=r.get("Grid_Size_X", r.get("Grid_Size", "")), wg=r.get("Workgroup_Size_X", ""))
    ctr = defaultdict(dict)
    if pmc:
        for r in csv.DictReader(open(pmc[0])):
            did = int(r["Dispatch_Id"])
            ctr[did][r["Counter_Name"]] = ctr[did].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            if did not in rows:
                rows[did] = dict(name=r["Kernel_Name"], dur=0, grid=r.get("Grid_Size", ""), wg=r.get("Workgroup_Size", ""))
    return rows, ctr


def main():
    for d in sys.argv[1:]:
        rows, ctr = load(d)
        ids = sorted(rows)
        last_relayout = max((i for i in ids if "strip_relayout" in rows[i]["name"]), default=-1)
        first_gemm = min((i for i in ids if "wna16_gemm_stream_kernel" in rows[i]["name"]), default=-1)
        acc = defaultdict(lambda: defaultdict(list))
        for i in ids:
            r = rows[i]
            if "wna16_gemm_stream_kernel" not in r["name"] or i < first_gemm:
                continue
            phase = "B" if i > last_relayout else "A"
            key = r["name"].split("(")[0].replace("void aphro::", "") + " grid " + str(r["grid"])
            acc[key][phase + ".dur_us"].append(r["dur"] / 1e3)
            for c, v in ctr.get(i, {}).items():
                acc[key][phase + "." + c].append(v)
        print(d)
        for key in sorted(acc):
            print("  " + key)
            names = sorted({k.split(".", 1)[1] for k in acc[key]})
            for n in names:
                a, b = acc[key].get("A." + n, []), acc[key].get("B." + n, [])
                ma = sum(a) / len(a) if a else float("nan")
                mb = sum(b) / len(b) if b else float("nan")
                print(f"    {n:>36}: A {ma:14.2f} (n={len(a)})   B {mb:14.2f} (n={len(b)})   B/A {mb / ma if ma else float('nan'):.3f}")


if __name__ == "__main__":
    main()
