# kernel trace of one bench configuration: bash tools/trace_bench_kernels.sh <outdir> <bench args...>
out=gpurun_out/$1; shift
mkdir -p $out
export TMPDIR=/tmp
timeout 280 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o t -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-prefill-info --no-ops-path --no-cpu-baseline "$@" > $out/line.json 2> $out/err.txt
find $out/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
rm -rf $out/prof
python - $out/kernel_stats.csv <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'aphro' in r['Name']:
        print(r['Name'][:90].ljust(90), r['Calls'], round(float(r['AverageNs']) / 1e3, 2), r['MinNs'])
P
