#!/bin/bash
# usage: tools/run_ablate.sh "K N M" variant-flag-sets...   (each arg = one set of -D flags, "" = baseline)
shape="$1"; shift
for v in "$@"; do
  name=$(echo "base $v" | tr -d '"')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -DABL_NAME="\"$name\"" $v tools/gemm_ablate.hip -o /tmp/abl 2>&1 | grep -E "error" ; timeout 60 /tmp/abl $shape
done
