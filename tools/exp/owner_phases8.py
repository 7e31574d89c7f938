#!/usr/bin/env python3
"""Where a workgroup of k_td_update_owner spends its time, finer (experiment build tools/exp/build/lib_owner_ph8.so):
clear | record loop (wave 0; wave 15) | terminal queue | barrier wait | flush | fallback-cache flush."""
import ctypes, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault('G2048_LIB', os.path.join(ROOT, 'tools', 'exp', 'build', 'lib_owner_ph8.so'))
pkg = importlib.import_module('2048_amd')
n, B = int(os.environ.get('N', 5)), 1 << 20
eng = pkg.Engine(B, n=n, seed=2048)
eng.init_weights(seed=7, scale=0.01)
alpha = 0.25 * eng.num_feat / (8.0 * B)
if os.environ.get('RULE', 'sum') == 'mean':
    eng.set_update_rule(1)
    alpha = 0.25
eng.td_steps(alpha, int(os.environ.get('STEPS', 900)))
eng.sync()
t = eng.debug_owner_plan().astype(np.int64)
ph = np.zeros(8192, np.uint64)
assert eng.lib.g2048_debug_owner_phases(ph.ctypes.data_as(ctypes.c_void_p)) == 0
ph = ph.astype(np.int64).reshape(1024, 8)[:len(t)]
start, end = t[:, 4], t[:, 5]
us = lambda a: a / 100.0
cols = {'clear': us(ph[:, 0] - start), 'loop w0': us(ph[:, 1] - ph[:, 0]), 'loop w15': us(ph[:, 6] - ph[:, 0]), 'queue': us(ph[:, 2] - ph[:, 1]),
        'barrier': us(ph[:, 3] - ph[:, 2]), 'flush': us(ph[:, 4] - ph[:, 3]), 'fb flush': us(end - ph[:, 4]), 'total': us(end - start)}
print(f'{len(t)} workgroups, makespan {us(end.max() - start.min()):.1f} us; microseconds, mean over the workgroups of a chunk:')
for v in sorted(set(t[:, 0])):
    for ch in sorted(set(t[t[:, 0] == v, 1])):
        m = (t[:, 0] == v) & (t[:, 1] == ch)
        print(f'  variant {v} chunk {ch:3d}: {m.sum():3d} wgs | ' + ' | '.join(f'{k} {c[m].mean():5.1f}' for k, c in cols.items()))
