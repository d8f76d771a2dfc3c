#!/bin/bash
# builds tools/bin/w4_lab (here, cross-compiled); run it on the GPU box: tools/bin/w4_lab
# (wna16_gemm_resident.hip rides along since round 6: the prompt GEMM asks it for the strip-major geometry)
set -e
cd "$(dirname "$0")/.."
bash tools/apply_lab_patches.sh > /dev/null      # the kernel sources WITH their lab branches: tools/bin/csrc_lab
mkdir -p tools/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-function -Wno-unused-variable -Wno-pass-failed $LAB_FLAGS \
  tools/w4_lab.hip tools/bin/csrc_lab/wna16_gemm_large.hip tools/bin/csrc_lab/wna16_gemm_resident.hip tools/bin/csrc_lab/runtime.hip -o tools/bin/w4_lab
echo built tools/bin/w4_lab
