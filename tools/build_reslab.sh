#!/bin/bash
# Lab builds of the resident W4A16 kernel for tools/reslab.hip: one small .so per flag set (runtime.o + the variant object).
#   tools/build_reslab.sh name1 "flags1" name2 "flags2" ...   -> tools/bin/lab_<name>.so ; also (re)builds tools/bin/reslab
set -e
cd "$(dirname "$0")/.."
CS=aphrodite_engine_amd/csrc
HIPCC=/opt/rocm/bin/hipcc
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-variable -Wno-pass-failed"
mkdir -p tools/bin
make -s -C $CS build/runtime.o
[ tools/bin/reslab -nt tools/reslab.hip ] || $HIPCC --offload-arch=gfx950 -O2 -std=c++17 tools/reslab.hip -o tools/bin/reslab -ldl &
SRC=${LAB_SRC:-wna16_gemm_resident.hip}
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  ( $HIPCC $BASE -DRES_LAB_SET $flags -c $CS/$SRC -o tools/bin/lab_$name.o &&
    $HIPCC --offload-arch=gfx950 -shared -fPIC -o tools/bin/lab_$name.so tools/bin/lab_$name.o $CS/build/runtime.o $LAB_LINK &&
    rm -f tools/bin/lab_$name.o && echo "built lab_$name.so" ) &
done
wait
