#!/bin/bash
# round 6: address-translation counters of the decode GEMMs before / after a release-and-rebuild of the layouts (separate --pmc passes)
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r6z
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r6z/trace -- python tools/prof_placement.py > gpurun_out/r6z/trace.log 2>&1)
(cd $R && timeout 200 rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum --kernel-trace --output-format csv -d gpurun_out/r6z/utcl1 -- python tools/prof_placement.py > gpurun_out/r6z/utcl1.log 2>&1)
(cd $R && timeout 200 rocprofv3 --pmc GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/r6z/utcl2 -- python tools/prof_placement.py > gpurun_out/r6z/utcl2.log 2>&1)
cd $R
python tools/prof_placement_reduce.py gpurun_out/r6z/trace gpurun_out/r6z/utcl1 gpurun_out/r6z/utcl2 > gpurun_out/r6z/summary.txt 2>&1
cat gpurun_out/r6z/summary.txt | head -80; tail -2 gpurun_out/r6z/utcl1.log gpurun_out/r6z/utcl2.log
find gpurun_out/r6z -name "*.csv" -size +3M -delete
