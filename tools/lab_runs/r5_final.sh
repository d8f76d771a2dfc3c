mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6) > gpurun_out/r5_gputests.log 2>&1
(timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/r5_bench_final.err | tail -1) > gpurun_out/r5_bench_final.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd $R && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5_bench_prof -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline --no-prefill-e2e > gpurun_out/r5_bench_prof.log 2>&1)
cd $R
tail -3 gpurun_out/r5_gputests.log; wc -c gpurun_out/r5_bench_final.json; ls gpurun_out/r5_bench_prof/*/ | head
