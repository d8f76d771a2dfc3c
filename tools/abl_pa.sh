#!/bin/bash
# builds lib variants with extra flags for paged_attention.hip and profiles the rope kernel
cd $GRAFT_REPO_ROOT/aphrodite_engine_amd/csrc
for v in "" "-DABL_PA_NOKV" "-DABL_PA_NOQ" "-DABL_PA_NOKV -DABL_PA_NOQ"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-variable -Wno-pass-failed -mllvm -amdgpu-mfma-vgpr-form=1 $v -c paged_attention.hip -o build/paged_attention.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libaphrodite_mi355x.so build/*.o
  cd /tmp; export TMPDIR=/tmp
  rm -rf /tmp/kt_x; timeout 90 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_x -o k -- python $GRAFT_REPO_ROOT/tools/prof_attn.py 1040 auto rope > /dev/null 2>&1
  echo "variant [$v]: $(grep paged_attention_kernel /tmp/kt_x/k_kernel_stats.csv | awk -F, '{print $(NF-5)}' | head -1) ns avg"
  cd $GRAFT_REPO_ROOT/aphrodite_engine_amd/csrc
done
