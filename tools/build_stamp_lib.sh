#!/bin/bash
# lab: the product library with the stream kernel stamped (tools/stream_stamps.patch applied to a scratch copy, -DAPHRO_STREAM_STAMPS) -> tools/bin/libaphro_stamps.so (never shipped)
set -e
cd "$(dirname "$0")/../aphrodite_engine_amd/csrc"
make -j8 > /dev/null
mkdir -p ../../tools/bin
cp wna16_gemm_resident.hip ../../tools/bin/wna16_gemm_resident_stamps.hip
patch -s ../../tools/bin/wna16_gemm_resident_stamps.hip ../../tools/stream_stamps.patch
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable -Wno-pass-failed \
  -mllvm -amdgpu-mfma-vgpr-form=1 -mllvm -amdgpu-kernarg-preload-count=14 -DAPHRO_STREAM_STAMPS $EXTRA_DEFS -I. -I../../include \
  -c ../../tools/bin/wna16_gemm_resident_stamps.hip -o ../../tools/bin/wna16_gemm_resident_stamps.o
objs=$(ls build/*.o | grep -v wna16_gemm_resident.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/${OUT_NAME:-libaphro_stamps.so} $objs ../../tools/bin/wna16_gemm_resident_stamps.o
