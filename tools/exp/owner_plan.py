#!/usr/bin/env python3
"""The LDS-owner plan in force and its workgroups' times (g2048_debug_owner_plan), n = 5 at 2^20 lanes after STEPS steps."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
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
start, end = t[:, 4], t[:, 5]
print(f'{len(t)} workgroups, makespan {(end.max() - start.min()) / 100.0:.1f} us, mean workgroup {((end - start) / 100.0).mean():.1f} us, start spread {(start.max() - start.min()) / 100.0:.1f} us')
for v in sorted(set(t[:, 0])):
    for ch in sorted(set(t[t[:, 0] == v, 1])):
        m = (t[:, 0] == v) & (t[:, 1] == ch)
        d = (end - start)[m] / 100.0
        print(f'  variant {v} chunk {ch:3d}: {m.sum():3d} wgs | mean {d.mean():5.1f} max {d.max():5.1f} us | sum {d.sum():7.1f}')
