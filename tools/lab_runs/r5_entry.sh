#!/bin/bash
# round 5: stream-kernel entry stamps, the resident/stream parity tests and a short bench after moving the placement divisions to the host
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 300 python tools/stream_stamps.py > gpurun_out/stream_stamps.log 2>&1
timeout 600 python -m pytest tests/test_resident_gpu.py -x -q -m gpu > gpurun_out/test_resident.log 2>&1
timeout 400 python bench.py --steps 30 --warmup 5 --no-extra-legs --no-cpu-baseline --no-prefill-e2e --no-prefill > gpurun_out/bench_entry.json 2> gpurun_out/bench_entry.err
tail -9 gpurun_out/stream_stamps.log; tail -3 gpurun_out/test_resident.log; tail -c 300 gpurun_out/bench_entry.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_entry.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline'])
PY
