#!/usr/bin/env python3
"""ms per TD step against the batch size (mean rule, aged boards): where fixed per-step costs show."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module('2048_amd')
n = int(os.environ.get('N', 5))
for logb in [int(x) for x in os.environ.get('LOGB', '16,17,18,19,20').split(',')]:
    B = 1 << logb
    eng = pkg.Engine(B, n=n, seed=2048)
    eng.init_weights(seed=7, scale=0.01)
    eng.set_update_rule(1)
    eng.td_steps(0.25, 600)
    eng.sync()
    t0 = time.perf_counter()
    eng.timer_start()
    eng.td_steps(0.25, 400)
    ms = eng.timer_stop()
    wall = time.perf_counter() - t0
    k = eng.td_steps_kernel_ms(0.25, 32)
    print(f'2^{logb} lanes: {ms / 400:.4f} ms/step (wall {wall / 400 * 1e3:.4f}), {B * 400 / ms * 1e3:.3e} board-steps/s; kernels ' + ' '.join(f'{x:.4f}' for x in k), flush=True)
    eng.close()
