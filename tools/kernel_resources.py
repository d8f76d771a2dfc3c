#!/usr/bin/env python3
"""Per-kernel resource table from the compiler's own metadata (no GPU needed): VGPRs, AGPRs, SGPRs, static LDS,
scratch (spills) and the occupancy they allow, for every kernel of every csrc/*.hip at the Makefile's flags.
usage: tools/kernel_resources.py [> profiles/rN_kernel_resources.txt]"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "aphrodite_engine_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-function", "-Wno-unused-variable", "-Wno-pass-failed"]
_VF, _KP = ["-mllvm", "-amdgpu-mfma-vgpr-form=1"], ["-mllvm", "-amdgpu-kernarg-preload-count=14"]
EXTRA = {"paged_attention.hip": _VF + _KP, "flash_attn.hip": _VF, "wna16_gemm_resident.hip": _VF + _KP,
         "fp8_gemm_resident.hip": _KP, "fp8_gemm_stream.hip": _KP, "wna16_gemm.hip": _KP}      # (the Makefile's per-file EXTRA)


def demangle(names):
    for tool in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "c++filt"):
        try:
            out = subprocess.run([tool], input="\n".join(names) + "\n", capture_output=True, text=True,
                                 check=True).stdout.splitlines()
            if len(out) == len(names):
                return dict(zip(names, out))
        except Exception:
            continue
    return {n: n for n in names}


def short_name(full):
    return re.sub(r"^void ", "", full.split("(")[0]).replace("aphro::", "")


def main():
    rows = []
    for src in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
        base = os.path.basename(src)
        with tempfile.NamedTemporaryFile(suffix=".s") as tf:
            cmd = [HIPCC] + FLAGS + EXTRA.get(base, []) + ["-S", "--cuda-device-only", "-o", tf.name, src]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                print(f"# {base}: compile failed\n{r.stderr[-400:]}", file=sys.stderr)
                continue
            txt = open(tf.name).read()
        for blk in re.split(r"\n  - \.agpr_count:", txt)[1:]:
            blk = ".agpr_count:" + blk

            def g(key):
                m = re.search(r"\." + key + r":\s+(\S+)", blk)
                return m.group(1) if m else "0"
            rows.append((base, g("name"), int(g("vgpr_count")), int(g("agpr_count")), int(g("sgpr_count")),
                         int(g("group_segment_fixed_size")), int(g("private_segment_fixed_size")),
                         int(g("max_flat_workgroup_size"))))
    names = demangle([r[1] for r in rows])
    print(f"{'file':22s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'lds B':>7s} {'scratch B':>9s} {'wg':>5s} waves/SIMD  kernel")
    for base, name, vg, ag, sg, lds, scr, wg in rows:
        regs = max(vg + ag, 1)
        occ = min(8, 512 // ((regs + 7) // 8 * 8))
        short = short_name(names[name])
        print(f"{base:22s} {vg:5d} {ag:5d} {sg:5d} {lds:7d} {scr:9d} {wg:5d} {occ:10d}  {short}")
    spilled = sorted({short_name(names[r[1]])[:70] for r in rows if r[6] > 0})
    print(f"\n{len(rows)} kernels, {len(spilled)} distinct with scratch (spills): " + ", ".join(spilled))


if __name__ == "__main__":
    main()
