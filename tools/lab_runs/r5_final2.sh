# round 5, second final pass (after the entry-cost work): FP8 + headline parity subsets, the default bench line, a kernel-trace of it
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_headline_gpu.py -m gpu -x -q -k "fp8 or scaled or headline or step" 2>&1 | tail -4) > gpurun_out/r5_fp8tests.log 2>&1
(timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/r5_bench_final2.err | tail -1) > gpurun_out/r5_bench_final2.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd $R && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5_bench_prof2 -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline --no-prefill-e2e > gpurun_out/r5_bench_prof2.log 2>&1)
cd $R
tail -3 gpurun_out/r5_fp8tests.log; wc -c gpurun_out/r5_bench_final2.json; ls gpurun_out/r5_bench_prof2/*/ | head
