# round 6 final pass after the one-copy weights: the whole GPU suite, smoke(), the driver bench line, a kernel trace of the same command
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r6final6
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r6final6/gputest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r6final6/gputest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6final6/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r6final6/smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6final6/bench.json 2> gpurun_out/r6final6/bench.err
cd /tmp
(cd $R && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6final6/prof -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline --no-prefill-e2e > gpurun_out/r6final6/prof.log 2>&1)
cd $R
# keep the stats summary only (the trace itself is large)
find gpurun_out/r6final6/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/r6final6/kernel_stats.csv \;
find gpurun_out/r6final6/prof -type f ! -name "*stats*" -delete
tail -4 gpurun_out/r6final6/gputest.log; tail -2 gpurun_out/r6final6/smoke.log
python - <<'P'
import json
d=json.loads(open('gpurun_out/r6final6/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_frac'])
for k,v in d.get('legs',{}).items(): print(k, v.get('ms_per_step'), v.get('value'), v.get('error'))
P
head -12 gpurun_out/r6final6/kernel_stats.csv | cut -c1-160
