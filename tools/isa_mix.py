#!/usr/bin/env python3
"""Instruction census (by mnemonic) of one kernel of the built library:  python3 tools/isa_mix.py [lib.so] 'k_td_play_hot<5, 512, 0>'"""
import collections, importlib.util, os, re, subprocess, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location('cc', os.path.join(HERE, 'check_codeobj.py'))
cc = importlib.util.module_from_spec(spec)
spec.loader.exec_module(cc)
so = sys.argv[1] if len(sys.argv) > 2 else os.path.join(cc.ROOT, '2048_amd', 'lib2048_hip.so')
pat = sys.argv[-1]
co = cc.extract_code_object(so, tempfile.mkdtemp())
for sym in cc.kernel_notes(co):
    base = sym[:-3] if sym.endswith('.kd') else sym
    dem = subprocess.check_output(['c++filt', base], text=True).strip()
    if pat not in dem:
        continue
    cnt = collections.Counter(text.split()[0] for _, text, _ in cc.disassemble(co, base))
    kinds = collections.Counter()
    for op, v in cnt.items():
        kinds['valu' if op.startswith('v_') else 'salu' if op.startswith('s_') else 'lds' if op.startswith('ds_') else 'vmem'] += v
    print(dem, sum(cnt.values()), dict(kinds))
    for k, v in cnt.most_common(int(os.environ.get('TOP', 40))):
        print(f'  {k:28s} {v}')
