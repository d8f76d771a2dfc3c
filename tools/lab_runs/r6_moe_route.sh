#!/bin/bash
# round 6: routing + alignment + gather in one launch (sparse MLP, decode)
mkdir -p gpurun_out/r6s
cd /root/repo
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_loader_gpu.py tests/test_headline_gpu.py -x -q -k "moe or mixtral or route or router" > gpurun_out/r6s/moe.log 2>&1; echo "rc=$?" >> gpurun_out/r6s/moe.log
timeout 900 python -m pytest tests/test_custom_ar_gpu.py -x -q -k "tp2_decode_fused" > gpurun_out/r6s/ar.log 2>&1; echo "rc=$?" >> gpurun_out/r6s/ar.log
B="--gpus 1 --steps 20 --warmup 5 --no-prefill-info --no-ops-path --no-cpu-baseline --no-extra-legs --no-prefill-e2e"
for rep in 1 2; do
  timeout 300 python bench.py $B --model mixtral-8x7b --sim-tp 4 > gpurun_out/r6s/cfg4_one_$rep.json 2> gpurun_out/r6s/cfg4_one_$rep.err
  APHRO_MOE_NO_ROUTE_ALIGN=1 timeout 300 python bench.py $B --model mixtral-8x7b --sim-tp 4 > gpurun_out/r6s/cfg4_sep_$rep.json 2> gpurun_out/r6s/cfg4_sep_$rep.err
done
timeout 300 python bench.py $B --model mixtral-8x7b > gpurun_out/r6s/mixtral_tp1.json 2> gpurun_out/r6s/mixtral_tp1.err
tail -n 4 gpurun_out/r6s/moe.log; tail -n 3 gpurun_out/r6s/ar.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r6s/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["ms_per_step"],4), round(d["value"]))
    except Exception as e: print(f, "ERR", e)
PY
