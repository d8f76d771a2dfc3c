#!/bin/bash
# round 5: PA fused-rope prologue (slab loads issued together) + placement divisions on the host: parity + bench
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "rope or paged_attention" > gpurun_out/test_pa.log 2>&1
timeout 900 python -m pytest tests/test_headline_gpu.py tests/test_resident_gpu.py -x -q -m gpu > gpurun_out/test_headline.log 2>&1
timeout 400 python bench.py --steps 30 --warmup 5 --no-extra-legs --no-cpu-baseline --no-prefill-e2e --no-prefill-info > gpurun_out/bench_entry.json 2> gpurun_out/bench_entry.err
tail -3 gpurun_out/test_pa.log; tail -3 gpurun_out/test_headline.log; tail -c 300 gpurun_out/bench_entry.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_entry.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline'])
print({k: v for k, v in d.items() if k in ('kernels', 'step_hbm', 'value_ops_path')})
PY
