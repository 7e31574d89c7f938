#!/usr/bin/env python3
"""Per-workgroup durations of the last k_td_update_owner launch on the bench workload (steady state)."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module('2048_amd')
n, B = int(os.environ.get('N', 5)), 1 << 20
eng = pkg.Engine(B, n=n, seed=2048)
eng.init_weights(seed=7, scale=0.01)
alpha = 0.25 * eng.num_feat / (8.0 * B)
if os.environ.get('RULE', 'sum') == 'mean':
    eng.set_update_rule(1)
    alpha = 0.25
eng.td_steps(alpha, int(os.environ.get("STEPS", 300)))
for rep in range(int(os.environ.get("REPS", 2))):
    a, b = eng.td_steps_profiled(alpha, 10)
    print(f'steady: play {a:.3f} update {b:.3f} ms')
    eng.sync()
    t = eng.debug_owner_plan().astype(np.int64)
    t0 = t[:, 4].min()
    start, end = (t[:, 4] - t0) / 100.0, (t[:, 5] - t0) / 100.0        # microseconds
    print(f'{len(t)} workgroups; first start 0, last start {start.max():.1f} us, last end {end.max():.1f} us; mean duration {(end - start).mean():.1f} us')
    for v in sorted(set(t[:, 0])):
        for ch in sorted(set(t[t[:, 0] == v, 1])):
            m = (t[:, 0] == v) & (t[:, 1] == ch)
            d = (end - start)[m]
            print(f'  variant {v} chunk {ch:3d}: {m.sum():3d} wgs (nparts {t[m, 3][0]}), duration min {d.min():6.1f} mean {d.mean():6.1f} max {d.max():6.1f} us, start max {start[m].max():5.1f}, end max {end[m].max():6.1f}')
    x = np.arange(len(t)) % 8
    print('  mean duration by blockIdx % 8 (XCD):', ' '.join(f'{(end - start)[x == k].mean():.1f}' for k in range(8)))
    print('  mean duration by record range (part * 8 // nparts):', ' '.join(f'{(end - start)[(t[:, 2] * 8 // t[:, 3]) == k].mean():.1f}' for k in range(8)))
