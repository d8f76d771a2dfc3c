"""Per-kernel statistics (and, with --seq, the launch sequence) from a rocprofv3 rocpd database:
    python tools/rocpd_stats.py <dir or .db> [--seq N]"""
import glob
import os
import sqlite3
import sys

path = sys.argv[1]
dbs = [path] if path.endswith(".db") else glob.glob(os.path.join(path, "**", "*.db"), recursive=True)
for db in dbs:
    con = sqlite3.connect(db)
    rows = con.execute("select name, start, end from kernels order by start").fetchall()
    if "--seq" in sys.argv:
        n = int(sys.argv[sys.argv.index("--seq") + 1])
        for name, s, e in rows[:n]:
            print(f"{(e - s) / 1e3:9.2f} us  {name[:110]}")
        continue
    agg = {}
    for name, s, e in rows:
        agg.setdefault(name, []).append(e - s)
    tot = sum(sum(v) for v in agg.values())
    print("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs")
    for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print(f"\"{name}\",{len(v)},{sum(v)},{sum(v) / len(v):.1f},{100.0 * sum(v) / tot:.3f},{min(v)},{max(v)}")
