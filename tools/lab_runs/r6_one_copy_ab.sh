#!/bin/bash
# round 6: the decode step with one resident copy of the int4 matrices against two (same box, alternating)
mkdir -p gpurun_out/r6v
for i in 1 2 3; do
  for mode in one two; do
    flag=""; [ $mode = two ] && flag="--two-copies"
    timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline --no-prefill-e2e $flag > gpurun_out/r6v/$mode$i.json 2> gpurun_out/r6v/$mode$i.err
    python - <<PY
import json
d=json.loads(open("gpurun_out/r6v/$mode$i.json").read().strip().splitlines()[-1])
print("$mode", $i, round(d["ms_per_step"],4), d["step_hbm"]["weight_bytes_resident"], {k:round(v["avg_us"],2) for k,v in d["roofline_all"].items()})
PY
  done
done
