#!/bin/bash
# round 5: same-box A/B of library variants on the configs[1] decode step (bench.py, short sections only)
# usage: r5_ab.sh name=path [name=path ...]   (two passes over the list, interleaved)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
: > gpurun_out/ab.jsonl
for pass in 1 2; do
  for spec in "$@"; do
    name="${spec%%=*}"; path="${spec#*=}"
    APHRODITE_MI355X_LIB="$PWD/$path" timeout 300 python bench.py --steps 200 --warmup 300 --no-extra-legs --no-cpu-baseline \
      --no-prefill-e2e --no-prefill-info --no-ops-path > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err
    python - "$name" "$pass" <<'PY' | tee -a gpurun_out/ab.jsonl
import json, sys
d = json.loads(open('gpurun_out/ab_tmp.json').read().strip().splitlines()[-1])
r = d['roofline_all']
print(json.dumps(dict(lib=sys.argv[1], run=int(sys.argv[2]), ms_per_step=round(d['ms_per_step'], 4),
                      us={k: round(v['avg_us'], 2) for k, v in r.items() if 'avg_us' in v})))
PY
  done
done
