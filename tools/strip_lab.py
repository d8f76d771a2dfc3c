#!/usr/bin/env python3
"""Resolve the kernel-lab preprocessor branches of a csrc/*.hip file for the PRODUCT tree and keep
what was removed as a patch under tools/lab_patches/ (cleaned -> lab form) that tools/apply_lab_patches.sh re-applies to a
scratch copy for the lab harnesses.

    python tools/strip_lab.py <file.hip> NAME[=value] ...      # NAME alone: undefined; NAME=v: defined to v

Handles `#ifdef / #ifndef / #if <expr> / #else / #endif` whose condition mentions ONLY the given names (anything else is left
alone), removes `#ifndef NAME / #define NAME default / #endif` default blocks of names given a value, and substitutes the
value for the remaining uses of those names."""
import os
import re
import subprocess
import sys


def evaluate(expr, defs):
    """True / False, or None when the expression mentions a name we were not told about."""
    e = re.sub(r"defined\s*\(\s*(\w+)\s*\)", lambda m: ("1" if defs.get(m.group(1)) is not None else "0") if m.group(1) in defs else "@", expr)
    names = set(re.findall(r"[A-Za-z_]\w*", e))
    if "@" in e or any(n not in defs for n in names):
        return None
    for n in names:
        e = re.sub(rf"\b{n}\b", str(defs[n] if defs[n] is not None else 0), e)
    e = e.replace("&&", " and ").replace("||", " or ").replace("!", " not ").replace(" not =", "!=")
    return bool(eval(e))


def main():
    path = sys.argv[1]
    defs = {}
    for a in sys.argv[2:]:
        n, _, v = a.partition("=")
        defs[n] = v if _ else None
    src = open(path).read().split("\n")
    out, stack = [], []          # stack entries: [decided (True/False/None), emitting_parent]
    i = 0
    while i < len(src):
        ln = src[i]
        st = ln.strip()
        emitting = all(s[0] is not False for s in stack if s[0] is not None) if stack else True
        m = re.match(r"#\s*(ifdef|ifndef|if|else|endif)\b\s*(.*?)\s*(//.*)?$", st)
        if m:
            kind, arg = m.group(1), m.group(2)
            if kind in ("ifdef", "ifndef"):
                name = arg.split()[0]
                if name in defs:
                    val = (defs[name] is not None) == (kind == "ifdef")
                    stack.append([val])
                    i += 1
                    continue
                stack.append([None])
            elif kind == "if":
                val = evaluate(arg, defs)
                stack.append([val])
                if val is not None:
                    i += 1
                    continue
            elif kind == "else":
                if stack[-1][0] is not None:
                    stack[-1][0] = not stack[-1][0]
                    i += 1
                    continue
            elif kind == "endif":
                top = stack.pop()
                if top[0] is not None:
                    i += 1
                    continue
        if emitting:
            out.append(ln)
        i += 1
    text = "\n".join(out)
    for n, v in defs.items():
        if v is not None:
            text = re.sub(rf"\b{n}\b", v, text)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = os.path.basename(path)
    lab_copy = os.path.join("/tmp", base + ".lab")
    open(lab_copy, "w").write("\n".join(src))
    open(path, "w").write(text)
    patch = subprocess.run(["diff", "-u", "--label", base, "--label", base, path, lab_copy], capture_output=True, text=True).stdout
    open(os.path.join(root, "tools", "lab_patches", base + ".patch"), "w").write(patch)
    print(f"{base}: {len(src)} -> {len(out)} lines, patch {len(patch.splitlines())} lines")


if __name__ == "__main__":
    main()
