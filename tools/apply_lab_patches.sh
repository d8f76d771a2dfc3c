#!/bin/bash
# The kernel sources WITH their lab branches (per-wave stamps, timing-only ablations, schedule variants: PA_LAB, FA_LAB, FA4_LAB,
# F8_LAB, ABL_*, LMH_*, FA_TRAIL, ...) -- removed from the product tree in round 6 -- as a scratch copy under tools/bin/csrc_lab/:
# the product sources + tools/lab_patches/*.patch.  The torch-free harnesses (tools/*.hip) and lab library builds include / compile
# from there:   bash tools/apply_lab_patches.sh && make -C tools/bin/csrc_lab LAB=1 EXTRA=-DPA_LAB ...
set -e
root="$(cd "$(dirname "$0")/.." && pwd)"
dst="$root/tools/bin/csrc_lab"
rm -rf "$dst" && mkdir -p "$dst"
cp "$root"/aphrodite_engine_amd/csrc/*.hip "$root"/aphrodite_engine_amd/csrc/*.h "$root"/aphrodite_engine_amd/csrc/Makefile "$dst"/
for p in "$root"/tools/lab_patches/*.patch; do
  f="$(basename "$p" .patch)"
  patch -s "$dst/$f" "$p"
done
# the copy sits two levels deeper than csrc/: point its includes / outputs back at the repo
sed -i 's|#include "../../include/aphrodite_mi355x.h"|#include "../../../include/aphrodite_mi355x.h"|' "$dst"/common.h
sed -i 's|\.\./\.\./include/aphrodite_mi355x\.h|../../../include/aphrodite_mi355x.h|; s|\.\./lib/|../../../aphrodite_engine_amd/lib/|g' "$dst"/Makefile
echo "lab sources in $dst"
