#!/bin/bash
# Ablation builds of csrc/wna16_gemm_mid.hip (MID_ABL bits: 1 no A loads, 2 no W loads, 4 no MFMA, 8 no dequant): one
# library per variant under tools/bin/, the other objects are the shipped ones.   tools/mid_ablate.sh 0 1 2 4 8 3 12 15
set -e
cd "$(dirname "$0")/.."
bash tools/apply_lab_patches.sh > /dev/null      # the kernel sources WITH their lab branches: tools/bin/csrc_lab
mkdir -p tools/bin
C=aphrodite_engine_amd/csrc
OTHERS=$(ls $C/build/*.o | grep -v wna16_gemm_mid.o)
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed -Wno-unused-variable -DMID_ABL=$v -c $C/wna16_gemm_mid.hip -o tools/bin/mid_abl$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/libmid_abl$v.so $OTHERS tools/bin/mid_abl$v.o
  rm tools/bin/mid_abl$v.o
done
