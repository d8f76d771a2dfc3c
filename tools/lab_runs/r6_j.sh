#!/bin/bash
mkdir -p gpurun_out/r6j
cd /root/repo
B="--gpus 1 --steps 20 --warmup 5 --no-prefill-info --no-ops-path --no-cpu-baseline --no-extra-legs --no-prefill-e2e"
L=/root/repo/aphrodite_engine_amd/lib
for ctx in 1024 2048 4096; do
 for v in old new; do
  case $v in old) lib=$L/libaphrodite_mi355x_old.so;; new) lib=$L/libaphrodite_mi355x.so;; esac
  APHRODITE_MI355X_LIB=$lib timeout 300 python bench.py $B --quant fp8ct --kv-cache-dtype fp8 --ctx $ctx > gpurun_out/r6j/fp8kv_${ctx}_${v}.json 2> gpurun_out/r6j/fp8kv_${ctx}_${v}.err
 done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r6j/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["ms_per_step"],4), round(d["value"]), round(d["roofline_all"]["paged_attention"]["avg_us"],2))
    except Exception as e: print(f, "ERR", e)
PY
B="--gpus 1 --steps 20 --warmup 5 --no-prefill-info --no-ops-path --no-cpu-baseline --no-extra-legs --no-prefill-e2e"
L=/root/repo/aphrodite_engine_amd/lib
for v in old new; do
  case $v in old) lib=$L/libaphrodite_mi355x_old.so;; new) lib=$L/libaphrodite_mi355x.so;; esac
  APHRODITE_MI355X_LIB=$lib timeout 300 python bench.py $B > gpurun_out/r6j/int4_${v}.json 2> gpurun_out/r6j/int4_${v}.err
  APHRODITE_MI355X_LIB=$lib timeout 300 python bench.py $B --quant fp8ct --kv-cache-dtype fp8 --ctx 8192 > gpurun_out/r6j/fp8kv_8192_${v}.json 2> gpurun_out/r6j/fp8kv_8192_${v}.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r6j/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["ms_per_step"],4), round(d["value"]), round(d["roofline_all"]["paged_attention"]["avg_us"],2))
    except Exception as e: print(f, "ERR", e)
PY
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "paged or attention_backend or split_kv or rope" 2>&1 | tail -3
