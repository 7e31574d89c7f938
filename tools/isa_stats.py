#!/usr/bin/env python3
"""Registers, LDS, scratch and an instruction census of the kernels in a hipcc --save-temps .s file (gfx950)."""
import re, sys, collections
s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else 'k_td_play'
for m in re.finditer(r'\n(_Z\w+):.*?\n(.*?)\.amdhsa_kernel \1(.*?)\.end_amdhsa_kernel', s, re.S):
    name, code, desc = m.group(1), m.group(2), m.group(3)
    if pat not in name:
        continue
    g = lambda k: (re.search(k + r'\s+(\S+)', desc) or [None, None])[1]
    census = collections.Counter()
    for line in code.splitlines():
        t = line.strip().split()
        if not t or t[0].startswith(('.', ';', '_Z')) or t[0].endswith(':'):
            continue
        op = t[0]
        kind = 'flat' if op.startswith('flat_') else 'global' if op.startswith('global_') else 'ds' if op.startswith('ds_') else 'scratch' if op.startswith('scratch_') else \
            'valu' if op.startswith('v_') else 'salu' if op.startswith('s_') else 'other'
        census[kind] += 1
    print(name, 'vgpr', g('next_free_vgpr'), 'sgpr', g('next_free_sgpr'), 'lds', g('group_segment_fixed_size'), 'scratch', g('private_segment_fixed_size'), dict(census))
