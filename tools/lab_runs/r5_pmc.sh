# round 5: HBM traffic per launch of the decode-step kernels (two separate PMC passes, no trace domains beside them)
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 110 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r5_pmc_fetch -- python tools/prof_step_kernels.py > gpurun_out/r5_pmc_fetch.log 2>&1)
(cd $R && timeout 110 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r5_pmc_write -- python tools/prof_step_kernels.py > gpurun_out/r5_pmc_write.log 2>&1)
cd $R
ls gpurun_out/r5_pmc_fetch/*/ gpurun_out/r5_pmc_write/*/ | head; tail -3 gpurun_out/r5_pmc_fetch.log
