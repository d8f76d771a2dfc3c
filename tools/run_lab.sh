cd $GRAFT_REPO_ROOT
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-prefill-info --no-ops-path --no-cpu-baseline --no-extra-legs --quant fp8ct"
show() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3), {k: round(v['avg_us'],2) for k,v in d['roofline_all'].items()})"; }
cp aphrodite_engine_amd/lib/libaphrodite_mi355x.so /tmp/keep.so
$B 2>/dev/null | show resident_A_d4
for d in 4 6 8; do
  cp tools/bin/lib_f8r_d$d.so aphrodite_engine_amd/lib/libaphrodite_mi355x.so
  $B 2>/dev/null | show streamA_d$d
done
python -m pytest tests/test_ops_gpu.py -x -q -k "fp8_gemm_resident" 2>&1 | tail -2
cp /tmp/keep.so aphrodite_engine_amd/lib/libaphrodite_mi355x.so
