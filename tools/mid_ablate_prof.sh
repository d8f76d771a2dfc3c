export PYTHONPATH=.; export TMPDIR=/tmp
for v in "$@"; do APHRODITE_MI355X_LIB=$PWD/tools/bin/libmid_abl$v.so timeout 60 rocprofv3 --kernel-trace -d gpurun_out/pm$v -- python tools/prof_mid.py 64 > /dev/null 2>&1; echo "== abl $v"; python tools/rocpd_stats.py gpurun_out/pm$v --seq 26 | sed -n "20,23p" | cut -c1-100; rm -rf gpurun_out/pm$v; done
