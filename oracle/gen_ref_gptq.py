#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY.  Build step of oracle/_ref/libaphro_ref_gptq.so.

Reads the reference's kernels/quantization/gptq/q_gemm.cu WHERE IT LIES and writes a host-compilable
view of it into oracle/_ref/ (git-ignored, never committed): every top-level definition is kept
verbatim EXCEPT the host-side launchers -- the functions that contain a `<<<...>>>` launch or touch
torch / ATen / cuBLAS / the CUDA runtime -- which g++ cannot parse and which oracle/ref_gptq_bind.cpp
replaces with an emulated launch of the same grid.  The kernels themselves (the arithmetic being
pinned) are compiled unmodified against oracle/cuda_host_shim/.

usage: gen_ref_gptq.py <reference q_gemm.cu> <output .inc>
"""
import re
import sys

FORBIDDEN = ("<<<", "torch::", "at::", "cublas", "cudaMalloc", "cudaStream_t", "hipblas")


def skip_trivia(src, j):
    """If a comment or a string / char literal starts at j, return the index just past it, else j."""
    if src.startswith("//", j):
        k = src.find("\n", j)
        return len(src) if k < 0 else k
    if src.startswith("/*", j):
        return src.index("*/", j) + 2
    if src[j] in "\"'":
        q = src[j]
        j += 1
        while src[j] != q:
            j += 2 if src[j] == "\\" else 1
        return j + 1
    return j


def scan(src):
    """Split the file into top-level items: ('def', a, b) for brace-bodied definitions, ('other', a, b) else."""
    i, n = 0, len(src)
    out = []
    decl_start = 0
    while i < n:
        j = skip_trivia(src, i)
        if j != i:
            i = j
            continue
        c = src[i]
        if c == "#" and src[src.rfind("\n", 0, i) + 1:i].strip() == "":
            # preprocessor line (with continuations): a boundary of its own
            j = i
            while True:
                k = src.find("\n", j)
                if k < 0:
                    k = n - 1
                    break
                if src[k - 1] == "\\":
                    j = k + 1
                    continue
                break
            out.append(("other", decl_start, k + 1))
            i = decl_start = k + 1
            continue
        if c == ";" or c == "}":          # end of a declaration / close of a namespace
            out.append(("other", decl_start, i + 1))
            i = decl_start = i + 1
            continue
        if c == "{":
            head = src[decl_start:i]
            if re.search(r"\bnamespace\b[^;{}()]*$", head) or re.search(r'extern\s+"C"\s*$', head):
                out.append(("other", decl_start, i + 1))
                i = decl_start = i + 1
                continue
            depth, j = 1, i + 1
            while depth:
                k = skip_trivia(src, j)
                if k != j:
                    j = k
                    continue
                if src[j] == "{":
                    depth += 1
                elif src[j] == "}":
                    depth -= 1
                j += 1
            k = j
            while k < n and src[k] in " \t\r\n":
                k += 1
            if k < n and src[k] == ";":   # struct / union / class definitions end with ';'
                j = k + 1
            out.append(("def", decl_start, j))
            i = decl_start = j
            continue
        i += 1
    if decl_start < n:
        out.append(("other", decl_start, n))
    return out


def main():
    ref, dst = sys.argv[1], sys.argv[2]
    src = open(ref).read()
    items = [(kind, src[a:b]) for kind, a, b in scan(src)]

    def name_of(text):
        m = re.search(r"([A-Za-z_]\w*)\s*\(", text)
        return m.group(1) if m else "?"

    drop = set()
    dropped = []
    changed = True
    while changed:          # host launchers, then whatever calls a dropped function
        changed = False
        for idx, (kind, text) in enumerate(items):
            if kind != "def" or idx in drop:
                continue
            calls_dropped = any(re.search(r"\b%s\s*\(" % re.escape(d), text) for d in dropped if d != "?")
            if any(t in text for t in FORBIDDEN) or calls_dropped:
                drop.add(idx)
                dropped.append(name_of(text))
                changed = True
    # keep line numbers aligned with the reference for compiler diagnostics
    kept = [("\n" * text.count("\n")) if idx in drop else text for idx, (kind, text) in enumerate(items)]
    with open(dst, "w") as f:
        f.write("// GENERATED at build time by oracle/gen_ref_gptq.py from %s -- do not commit.\n" % ref)
        f.write("// host launchers removed: %s\n" % ", ".join(dropped))
        f.write("#line 1 \"%s\"\n" % ref)
        f.write("".join(kept))
    print("[gen_ref_gptq] dropped %d host launchers: %s" % (len(dropped), ", ".join(dropped)))


if __name__ == "__main__":
    main()
