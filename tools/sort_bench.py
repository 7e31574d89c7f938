#!/usr/bin/env python3
"""Whole-step time of the bench workload against the lane re-ordering period (g2048_set_lane_sort)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module('2048_amd')
n, B = int(os.environ.get('N', 5)), 1 << 20
rule = os.environ.get('RULE', 'sum')
for every in [int(x) for x in os.environ.get('EVERY', '0,4,8,16,32').split(',')]:
    eng = pkg.Engine(B, n=n, seed=2048)
    eng.init_weights(seed=7, scale=0.01)
    alpha = 0.25 * eng.num_feat / (8.0 * B)
    if rule == 'mean':
        eng.set_update_rule(1)
        alpha = 0.25
    eng.set_lane_sort(every)
    eng.td_steps(alpha, int(os.environ.get('AGE', 320)))
    eng.sync()
    ts = []
    for rep in range(3):
        eng.timer_start()
        eng.td_steps(alpha, 192)
        ts.append(eng.timer_stop() / 192)
    k = eng.td_steps_kernel_ms(alpha, 32)
    print(f'sort every {every:3d}: ms/step {sorted(ts)[1]:.4f}   kernels (play incl. sort, owner, tail, apply) ' + ' '.join(f'{x:.4f}' for x in k), flush=True)
    eng.close()
