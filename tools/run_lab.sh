cd $GRAFT_REPO_ROOT
python -m pytest tests/test_resident_gpu.py tests/test_headline_gpu.py -x -q 2>&1 | tail -3
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-prefill-info --no-ops-path --no-cpu-baseline --no-extra-legs"
show() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],4), {k: round(v['avg_us'],2) for k,v in d['roofline_all'].items()})"; }
$B 2>/dev/null | show stream
APHRO_WNA16_STREAM=0 $B 2>/dev/null | show twopass
$B 2>/dev/null | show stream
APHRO_WNA16_STREAM=0 $B 2>/dev/null | show twopass
$B --quant awq 2>/dev/null | show awq_stream
