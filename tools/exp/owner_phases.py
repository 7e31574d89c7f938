#!/usr/bin/env python3
"""Where a workgroup of k_td_update_owner spends its time (experiment build: tools/exp/r04_owner_phase_stamps.patch ->
tools/exp/build/lib2048_hip_phases.so, loaded through G2048_LIB): clear | record loop + terminal queue | barrier | flush."""
import ctypes, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault('G2048_LIB', os.path.join(ROOT, 'tools', 'exp', 'build', 'lib2048_hip_phases.so'))
pkg = importlib.import_module('2048_amd')
n, B = int(os.environ.get('N', 5)), 1 << 20
eng = pkg.Engine(B, n=n, seed=2048)
eng.init_weights(seed=7, scale=0.01)
alpha = 0.25 * eng.num_feat / (8.0 * B)
if os.environ.get('RULE', 'sum') == 'mean':
    eng.set_update_rule(1)
    alpha = 0.25
eng.td_steps(alpha, int(os.environ.get('STEPS', 300)))
eng.sync()
t = eng.debug_owner_plan().astype(np.int64)
ph = np.zeros(4096, np.uint64)
assert eng.lib.g2048_debug_owner_phases(ph.ctypes.data_as(ctypes.c_void_p)) == 0
ph = ph.astype(np.int64).reshape(1024, 4)[:len(t)]
start, end = t[:, 4], t[:, 5]
clear, loop, bar, flush = (ph[:, 0] - start) / 100.0, (ph[:, 1] - ph[:, 0]) / 100.0, (ph[:, 2] - ph[:, 1]) / 100.0, (end - ph[:, 2]) / 100.0
print(f'{len(t)} workgroups, makespan {(end.max() - start.min()) / 100.0:.1f} us; microseconds (thread 0 of every workgroup):')
for v in sorted(set(t[:, 0])):
    for ch in sorted(set(t[t[:, 0] == v, 1])):
        m = (t[:, 0] == v) & (t[:, 1] == ch)
        print(f'  variant {v} chunk {ch:3d}: {m.sum():3d} wgs | clear {clear[m].mean():5.1f} | loop {loop[m].mean():5.1f} (max {loop[m].max():5.1f}) | barrier {bar[m].mean():5.1f} (max {bar[m].max():5.1f}) | flush {flush[m].mean():5.1f} (max {flush[m].max():5.1f}) | total {((end - start)[m] / 100.0).mean():5.1f}')
