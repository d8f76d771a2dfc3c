cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/ -x -q -m gpu -k "norm or fused_decode or headline" 2>&1 | tail -3
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-prefill-info --no-ops-path --no-cpu-baseline --no-extra-legs"
$B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new  ', d['value'], d['ms_per_step'])"
APHRO_NORM_GENERIC=1 $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('old  ', d['value'], d['ms_per_step'])"
$B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new  ', d['value'], d['ms_per_step'])"
