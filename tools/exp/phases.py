"""Mean time per wave and phase of k_td_play.  Needs an instrumented build (not kept in the tree): wall_clock64() stamps
at the phase boundaries of k_td_play accumulated per wave slot in a __device__ array and an extra entry point
g2048_debug_phases(out[8], reset); results are quoted in DESIGN.md section 4."""
import importlib, os, sys, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module('2048_amd')
lib = importlib.import_module('2048_amd._lib').load()
n, B = 5, 1 << 20
eng = pkg.Engine(B, n=n, seed=2048)
eng.init_weights(seed=7, scale=0.01)
alpha = 0.25 * eng.num_feat / (8.0 * B)
eng.td_steps(alpha, 300); eng.sync()
out = (ctypes.c_ulonglong * 8)()
lib.g2048_debug_phases(out, 1)
steps = 20
a, b = eng.td_steps_profiled(alpha, steps); eng.sync()
lib.g2048_debug_phases(out, 0)
waves = B / 64 * steps
names = ['stats init+barrier', 'state loads', 'all_moves', 'choose (gathers)', 'pick/spawn/over/index stores', 'reset + stores', 'stats flush']
print(f'play {a:.3f} ms update {b:.3f} ms; mean microseconds per wave by phase:')
tot = 0
for k in range(7):
    us = out[k] / waves / 100.0
    tot += us
    print(f'  {names[k]:34s} {us:8.2f} us')
print(f'  total per wave-iteration {tot:.2f} us')
