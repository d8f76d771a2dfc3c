#!/bin/bash
# tools/isa_of.sh <file.hip> <kernel-name-substring> [extra flags]: compile one csrc file with -save-temps and print the ISA of the first kernel matching
set -e
cd /root/repo/aphrodite_engine_amd/csrc
f=$1; k=$2; shift 2
rm -rf /tmp/isa && mkdir -p /tmp/isa && cd /tmp/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-variable -Wno-pass-failed -I/root/repo/aphrodite_engine_amd/csrc "$@" -save-temps -c /root/repo/aphrodite_engine_amd/csrc/$f -o /tmp/isa/x.o 2>&1 | grep -E "error" || true
S=$(ls /tmp/isa/*gfx950.s | head -1)
name=$(grep -o "^_Z[A-Za-z0-9_]*${k}[A-Za-z0-9_]*:" $S | head -1 | tr -d ':')
awk -v n="$name" '$0 ~ "^"n":" {p=1} p {print} p && /\.end_amdhsa_kernel/ {exit}' $S > /tmp/isa/k.s
echo "$name: $(wc -l < /tmp/isa/k.s) lines"
grep -E "\.amdhsa_next_free_vgpr|\.amdhsa_accum_offset|scratch_en|private_segment_fixed" /tmp/isa/k.s | head
