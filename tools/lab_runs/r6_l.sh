#!/bin/bash
mkdir -p gpurun_out/r6l
cd /root/repo
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6l/bench.json 2> gpurun_out/r6l/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6l/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_frac'])
for k,v in d.get('legs',{}).items(): print(k, v.get('ms_per_step'), v.get('value'), v.get('error'))
print(json.dumps(d.get('prefill_e2e'))[:1200])
PY
