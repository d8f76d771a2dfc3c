#!/bin/bash
# builds tools/bin/gemm8_lab (here, cross-compiled) ; run it on the GPU box:  tools/bin/gemm8_lab [quick]
set -e
cd "$(dirname "$0")/.."
bash tools/apply_lab_patches.sh > /dev/null      # the kernel sources WITH their lab branches: tools/bin/csrc_lab
mkdir -p tools/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-function -Wno-unused-variable -Wno-pass-failed -DF8_LAB $LAB_FLAGS \
  tools/gemm8_lab.hip tools/bin/csrc_lab/fp8_gemm_large.hip tools/bin/csrc_lab/runtime.hip -o tools/bin/gemm8_lab
echo built tools/bin/gemm8_lab
