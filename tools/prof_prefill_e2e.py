#!/usr/bin/env python3
"""One 8192-token int4 prompt through the model (bench.py's prefill_e2e, its 2 warm-up + 3 timed passes), for
`rocprofv3 --kernel-trace --stats`: where its ~102 ms go.  `--reduce <kernel_stats.csv>`: per-kernel us per layer and pass."""
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--reduce":
        rows = list(csv.DictReader(open(sys.argv[2])))
        tot = sum(float(r["TotalDurationNs"]) for r in rows)
        print(f"all kernels: {tot / 5 / 1e6:.2f} ms per pass (5 passes traced)")
        for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:16]:
            n = r["Name"].split("(")[0].replace("void aphro::", "")[:70]
            print(f"  {n:<70} calls {int(r['Calls']):>5}  {float(r['TotalDurationNs']) / 5 / 1e6:7.2f} ms/pass  {float(r['AverageNs']) / 1e3:8.1f} us avg")
    else:
        import bench
        which = tuple(sys.argv[1].split(",")) if len(sys.argv) > 1 else ("int4",)
        out = bench.prefill_e2e_section(which=which, library=False)
        print({k: round(v["ms"], 2) for k, v in out.items()})
