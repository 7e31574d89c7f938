#!/usr/bin/env python3
"""Per-step kernel times of the TD loop (diagnostic): step, ms_play, ms_update, episodes finished so far."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module('2048_amd')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 600
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 20
mode = int(sys.argv[4]) if len(sys.argv) > 4 else 1
eng = pkg.Engine(B, n=n, seed=2048)
eng.init_weights(seed=7, scale=0.01)
eng.set_update_mode(mode)
alpha = 0.25 * eng.num_feat / (8.0 * B)
prev = 0
for t in range(steps):
    a, b = eng.td_steps_profiled(alpha, 1)
    if t % 10 == 0 or b > 3.0:
        ep = eng.stats()['episodes']
        print(f'step {t:4d} play {a:7.3f} ms update {b:8.3f} ms finished {ep - prev}', flush=True)
        prev = ep
