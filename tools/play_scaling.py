#!/usr/bin/env python3
"""k_td_play / update time against the lane count (is there a fixed part?)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module('2048_amd')
n = int(os.environ.get('N', 5))
for lb in (14, 16, 17, 18, 19, 20, 21, 22):
    B = 1 << lb
    eng = pkg.Engine(B, n=n, seed=2048)
    eng.init_weights(seed=7, scale=0.01)
    alpha = 0.25 * eng.num_feat / (8.0 * B)
    eng.td_steps(alpha, 200)
    a, b = eng.td_steps_profiled(alpha, 20)
    print(f'B=2^{lb}: play {a * 1e3:8.1f} us ({a * 1e6 / B:6.3f} ns/lane)  update {b * 1e3:8.1f} us', flush=True)
    eng.close()
