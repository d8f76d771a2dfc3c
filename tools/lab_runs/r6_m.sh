#!/bin/bash
mkdir -p gpurun_out/r6m
cd /root/repo
B="--gpus 1 --steps 20 --warmup 5 --no-prefill-info --no-ops-path --no-cpu-baseline --no-extra-legs --no-prefill-e2e"
L=/root/repo/aphrodite_engine_amd/lib
APHRODITE_MI355X_LIB=$L/libaphrodite_mi355x_db.so timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "paged or attention_backend or split_kv or rope" 2>&1 | tail -3
for rep in 1 2; do
 for v in base db; do
  case $v in base) lib=$L/libaphrodite_mi355x.so;; db) lib=$L/libaphrodite_mi355x_db.so;; esac
  APHRODITE_MI355X_LIB=$lib timeout 300 python bench.py $B > gpurun_out/r6m/int4_${v}_$rep.json 2> gpurun_out/r6m/int4_${v}_$rep.err
 done
done
for v in base db; do
  case $v in base) lib=$L/libaphrodite_mi355x.so;; db) lib=$L/libaphrodite_mi355x_db.so;; esac
  APHRODITE_MI355X_LIB=$lib timeout 300 python bench.py $B --quant fp8ct --kv-cache-dtype fp8 --ctx 8192 > gpurun_out/r6m/cfg2_${v}.json 2> gpurun_out/r6m/cfg2_${v}.err
  APHRODITE_MI355X_LIB=$lib timeout 300 python bench.py $B --quant fp8ct --kv-cache-dtype fp8 --ctx 1024 > gpurun_out/r6m/fp8kv1k_${v}.json 2> gpurun_out/r6m/fp8kv1k_${v}.err
  APHRODITE_MI355X_LIB=$lib timeout 300 python bench.py $B --ctx 8192 > gpurun_out/r6m/int4c8k_${v}.json 2> gpurun_out/r6m/int4c8k_${v}.err
  APHRODITE_MI355X_LIB=$lib timeout 300 python bench.py $B --ragged > gpurun_out/r6m/ragged_${v}.json 2> gpurun_out/r6m/ragged_${v}.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r6m/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["ms_per_step"],4), round(d["value"]), round(d["roofline_all"]["paged_attention"]["avg_us"],2))
    except Exception as e: print(f, "ERR", e)
PY
