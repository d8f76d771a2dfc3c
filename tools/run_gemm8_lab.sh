#!/bin/bash
# builds tools/bin/gemm8_lab (here, cross-compiled) ; run it on the GPU box:  tools/bin/gemm8_lab [quick]
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-function -Wno-unused-variable -Wno-pass-failed -DF8_LAB $LAB_FLAGS \
  tools/gemm8_lab.hip aphrodite_engine_amd/csrc/fp8_gemm_large.hip aphrodite_engine_amd/csrc/runtime.hip -o tools/bin/gemm8_lab
echo built tools/bin/gemm8_lab
