#!/bin/bash
# round 5: translation-vs-latency probe + the bench's vs_library section
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 120 tools/bin/tlb_probe 96 60 > gpurun_out/tlb_probe_60.jsonl 2>&1
timeout 120 tools/bin/tlb_probe 192 12 > gpurun_out/tlb_probe_12.jsonl 2>&1
timeout 400 python bench.py --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline --no-prefill-e2e > gpurun_out/bench_vs_library.json 2> gpurun_out/bench_vs_library.err
tail -c 600 gpurun_out/bench_vs_library.err
cat gpurun_out/tlb_probe_60.jsonl gpurun_out/tlb_probe_12.jsonl
