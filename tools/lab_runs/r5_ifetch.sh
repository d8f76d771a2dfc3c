#!/bin/bash
# round 5: instruction-fetch probe + stream-kernel stamps
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 120 tools/bin/ifetch_probe > gpurun_out/ifetch_probe.jsonl 2>&1
timeout 300 python tools/stream_stamps.py > gpurun_out/stream_stamps.log 2>&1
cat gpurun_out/ifetch_probe.jsonl; tail -20 gpurun_out/stream_stamps.log
