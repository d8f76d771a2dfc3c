#!/bin/bash
# round 6: FETCH_SIZE of the prompt-sized W4A16 GEMM on [K/8, N] and on the strip-major copy (one --pmc pass, kernel trace beside it)
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r6sf
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r6sf/fetch -- python tools/prof_strip_fetch.py > gpurun_out/r6sf/fetch.log 2>&1)
cd $R
python tools/prof_strip_fetch.py --reduce gpurun_out/r6sf/fetch > gpurun_out/r6sf/summary.txt 2>&1
cat gpurun_out/r6sf/summary.txt; tail -n 2 gpurun_out/r6sf/fetch.log
find gpurun_out/r6sf -name "*.csv" -size +3M -delete
