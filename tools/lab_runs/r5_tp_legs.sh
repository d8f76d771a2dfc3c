mkdir -p gpurun_out
(timeout 500 python -m pytest tests/test_custom_ar_gpu.py -x -q 2>&1 | tail -8) > gpurun_out/r5_ar_tests2.log 2>&1
(timeout 100 python tools/ar_norm_bench.py 2>&1 | grep world) > gpurun_out/r5_ar_norm_bench.txt
B="--steps 20 --warmup 5 --no-cpu-baseline --no-prefill-info --no-ops-path --no-prefill-e2e"
C3="--model llama3-70b --quant awq --batch 64 --sim-tp 8"
C4="--model mixtral-8x7b --sim-tp 4"
run() { name=$1; shift; (timeout 150 env "$@" 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],3), 'ms', {k:round(v['avg_us'],2) for k,v in d.get('roofline_all',{}).items()})") >> gpurun_out/r5_tp_legs.txt 2>&1; }
rm -f gpurun_out/r5_tp_legs.txt
run cfg3_fused_prefetch APHRO_AR_PREFETCH=1 python bench.py $B $C3
run cfg3_fused APHRO_X=1 python bench.py $B $C3
run cfg3_unfused APHRO_NO_FUSED_AR_NORM=1 python bench.py $B $C3
run cfg3_stub APHRO_X=1 python bench.py $B $C3 --sim-ar-stub
run cfg4_fused_prefetch APHRO_AR_PREFETCH=1 python bench.py $B $C4
run cfg4_fused APHRO_X=1 python bench.py $B $C4
run cfg4_unfused APHRO_NO_FUSED_AR_NORM=1 python bench.py $B $C4
run cfg4_stub APHRO_X=1 python bench.py $B $C4 --sim-ar-stub
tail -3 gpurun_out/r5_ar_tests2.log; cat gpurun_out/r5_ar_norm_bench.txt; cat gpurun_out/r5_tp_legs.txt
