cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_bench_default.json 2> gpurun_out/r4_bench_default.err
echo bench rc=$?
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r4_bench_default.json').read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "step_frac", d["roofline"]["step_frac"], "dom", d["roofline"]["kernel"], d["roofline"]["frac"])
print({k: round(v["avg_us"],2) for k,v in d["roofline_all"].items()})
print("ops_path", d.get("value_ops_path"))
for k,v in d.get("legs",{}).items():
    print(k, {kk: vv for kk,vv in v.items() if kk in ("value","ms_per_step","error")})
PY
python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline --no-prefill-info 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('again', d['value'], d['ms_per_step'], d.get('value_ops_path'))"
python bench.py --gpus 1 --steps 64 --warmup 8 --no-extra-legs --no-cpu-baseline --no-prefill-info --no-ops-path 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('again64', d['value'], d['ms_per_step'], d.get('value_ops_path'))"
