cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_ops_gpu.py -x -q -k "fp8_gemm_resident" 2>&1 | tail -4
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-prefill-info --no-ops-path --no-cpu-baseline --no-extra-legs"
show() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3), {k: round(v['avg_us'],2) for k,v in d['roofline_all'].items()})"; }
$B --quant fp8ct 2>/dev/null | show new_fp8ct
APHRO_DECODE_NO_FP8_RESIDENT=1 $B --quant fp8ct 2>/dev/null | show old_fp8ct
$B --quant fp8ct 2>/dev/null | show new_fp8ct
$B --quant fp8ct --kv-cache-dtype fp8 --ctx 8192 2>/dev/null | show new_cfg2
APHRO_DECODE_NO_FP8_RESIDENT=1 $B --quant fp8ct --kv-cache-dtype fp8 --ctx 8192 2>/dev/null | show old_cfg2
$B --quant fp8ct --act-scheme static 2>/dev/null | show new_static
APHRO_DECODE_NO_FP8_RESIDENT=1 $B --quant fp8ct --act-scheme static 2>/dev/null | show old_static
python -m pytest tests/test_headline_gpu.py -x -q -k "config2" 2>&1 | tail -3
