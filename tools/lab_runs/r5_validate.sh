#!/bin/bash
# round 5: same-box A/B (head / stream-kernel preload only / everything) then the whole GPU suite
cd "$GRAFT_REPO_ROOT"
bash tools/lab_runs/r5_ab.sh head=tools/bin/lib_head.so preload=tools/bin/lib_preload.so new=aphrodite_engine_amd/lib/libaphrodite_mi355x.so
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_suite.log 2>&1
tail -5 gpurun_out/gpu_suite.log
