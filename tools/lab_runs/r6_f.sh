#!/bin/bash
# round 6 (f): FP8 TP fused all-reduce + norm + quant
mkdir -p gpurun_out/r6f
cd /root/repo
timeout 1500 python -m pytest tests/test_custom_ar_gpu.py -x -q > gpurun_out/r6f/ar.log 2>&1; echo "rc=$?" >> gpurun_out/r6f/ar.log
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_fp8_diet_gpu.py -x -q -k "tp2 or fp8 or norm" > gpurun_out/r6f/ops.log 2>&1; echo "rc=$?" >> gpurun_out/r6f/ops.log
B="--gpus 1 --steps 20 --warmup 5 --no-prefill-info --no-ops-path --no-cpu-baseline --no-extra-legs --no-prefill-e2e"
for rep in 1 2; do
  timeout 300 python bench.py $B --quant fp8ct --sim-tp 2 > gpurun_out/r6f/fp8_tp2_fused_$rep.json 2> gpurun_out/r6f/fp8_tp2_fused_$rep.err
  APHRO_NO_FUSED_AR_NORM=1 timeout 300 python bench.py $B --quant fp8ct --sim-tp 2 > gpurun_out/r6f/fp8_tp2_two_$rep.json 2> gpurun_out/r6f/fp8_tp2_two_$rep.err
done
tail -3 gpurun_out/r6f/ar.log gpurun_out/r6f/ops.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r6f/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["ms_per_step"], d["value"])
    except Exception as e: print(f, "ERR", e)
PY
