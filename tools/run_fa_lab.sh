#!/bin/bash
# tools/run_fa_lab.sh [extra hipcc flags]: build the attention lab here (no GPU needed); run tools/bin/fa_lab [T] [dtype] on the GPU box
set -e
cd "$(dirname "$0")/.."
bash tools/apply_lab_patches.sh > /dev/null      # the kernel sources WITH their lab branches: tools/bin/csrc_lab
mkdir -p tools/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Wno-pass-failed "$@" tools/fa_lab.hip -o tools/bin/fa_lab
