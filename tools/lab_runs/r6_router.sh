#!/bin/bash
# round 6: all-reduce + norm + router in one launch (sparse-MLP layers under TP)
mkdir -p gpurun_out/r6r
cd /root/repo
timeout 1500 python -m pytest tests/test_custom_ar_gpu.py -x -q -k "router or tp2_decode_fused or loopback" > gpurun_out/r6r/ar.log 2>&1; echo "rc=$?" >> gpurun_out/r6r/ar.log
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "router or moe or mixtral" > gpurun_out/r6r/moe.log 2>&1; echo "rc=$?" >> gpurun_out/r6r/moe.log
B="--gpus 1 --steps 20 --warmup 5 --no-prefill-info --no-ops-path --no-cpu-baseline --no-extra-legs --no-prefill-e2e"
for rep in 1 2; do
  timeout 300 python bench.py $B --model mixtral-8x7b --sim-tp 4 > gpurun_out/r6r/cfg4_fused_$rep.json 2> gpurun_out/r6r/cfg4_fused_$rep.err
  APHRO_NO_FUSED_AR_NORM=1 timeout 300 python bench.py $B --model mixtral-8x7b --sim-tp 4 > gpurun_out/r6r/cfg4_two_$rep.json 2> gpurun_out/r6r/cfg4_two_$rep.err
done
tail -n 4 gpurun_out/r6r/ar.log; tail -n 4 gpurun_out/r6r/moe.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r6r/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["ms_per_step"],4), round(d["value"]))
    except Exception as e: print(f, "ERR", e)
PY
