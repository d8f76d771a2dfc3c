mkdir -p gpurun_out/fp8ct_trace
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 280 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fp8ct_trace/prof -o t -- python bench.py --gpus 1 --steps 20 --warmup 5 --quant fp8ct --kv-cache-dtype fp8 --ctx 8192 --no-cpu-baseline --no-prefill-info --no-ops-path > gpurun_out/fp8ct_trace/line.json 2> gpurun_out/fp8ct_trace/err.txt
find gpurun_out/fp8ct_trace/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/fp8ct_trace/kernel_stats.csv
rm -rf gpurun_out/fp8ct_trace/prof
head -30 gpurun_out/fp8ct_trace/kernel_stats.csv | cut -c1-200
tail -3 gpurun_out/fp8ct_trace/err.txt
