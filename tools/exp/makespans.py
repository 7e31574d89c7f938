#!/usr/bin/env python3
"""Last makespans of k_td_update_owner (G2048_DEBUG_PLAN feedback lines) of an n = 5 context trained under the mean rule for STEPS steps."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ['G2048_DEBUG_PLAN'] = '1'
pkg = importlib.import_module('2048_amd')
B = 1 << 20
eng = pkg.Engine(B, n=5, seed=2048)
eng.init_weights(seed=7, scale=0.01)
eng.set_update_rule(1)
eng.td_steps(0.25, int(os.environ.get('STEPS', 4500)))
eng.sync()
ts = []
for rep in range(3):
    eng.timer_start()
    eng.td_steps(0.25, 192)
    ts.append(eng.timer_stop() / 192)
k = eng.td_steps_kernel_ms(0.25, 32)
print('ms/step', sorted(ts)[1], 'kernels', [round(float(x), 4) for x in k], file=sys.stderr)
