#!/usr/bin/env python3
"""Instruction mix of the kernels in a hipcc -S --cuda-device-only listing: counts per class (VALU / MFMA / VMEM / LDS /
SALU / waitcnt / accvgpr moves) and the most frequent VALU opcodes.  usage: isa_mix.py file.s [substring-of-kernel-name]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r'\n(_Z\S+):\s*; @\S+\n(.*?)\n\.Lfunc_end', s, flags=re.S):
    name, body = m.group(1), m.group(2)
    if pat not in name:
        continue
    cnt = collections.Counter()
    valu = collections.Counter()
    for line in body.split('\n'):
        line = line.strip()
        if not line or line[0] in '.;/' or line.endswith(':'):
            continue
        op = line.split()[0]
        if op.startswith('v_mfma'):
            k = 'mfma'
        elif op.startswith('v_accvgpr'):
            k = 'accvgpr_mov'
        elif op.startswith('v_'):
            k = 'valu'
            valu[op] += 1
        elif op.startswith(('buffer_', 'global_', 'flat_', 'scratch_')):
            k = 'vmem'
        elif op.startswith('ds_'):
            k = 'lds'
        elif op.startswith('s_waitcnt'):
            k = 'waitcnt'
        elif op.startswith('s_nop'):
            k = 's_nop'
        elif op.startswith('s_'):
            k = 'salu'
        else:
            k = 'other'
        cnt[k] += 1
        cnt['total'] += 1
    print(name)
    print('  ', dict(cnt))
    print('  ', valu.most_common(12))
