#!/usr/bin/env python3
"""Per-iteration clock stamps of the record loop of k_td_update_owner (experiment build: tools/exp/r04_owner_iteration_stamps.patch
-> tools/exp/build/lib_owner_iter.so, through G2048_LIB): workgroups 0-3 and 200-203, lane 0 of every wave, wall_clock64 (100 MHz)
at the top of each loop iteration and once after the loop."""
import ctypes, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault('G2048_LIB', os.path.join(ROOT, 'tools', 'exp', 'build', 'lib_owner_iter.so'))
pkg = importlib.import_module('2048_amd')
n, B = 5, 1 << 20
eng = pkg.Engine(B, n=n, seed=2048)
eng.init_weights(seed=7, scale=0.01)
eng.td_steps(0.25 * eng.num_feat / (8.0 * B), int(os.environ.get('STEPS', 300)))
eng.sync()
t = eng.debug_owner_plan().astype(np.int64)
clk = np.zeros(8 * 16 * 40, np.uint64)
assert eng.lib.g2048_debug_iter_clk(clk.ctypes.data_as(ctypes.c_void_p)) == 0
clk = clk.astype(np.int64).reshape(8, 16, 40)
for w, wg in enumerate((0, 1, 2, 3, 200, 201, 202, 203)):
    if wg >= len(t):
        continue
    start, end = t[wg, 4], t[wg, 5]
    print(f'workgroup {wg}: plan row {t[wg, :4].tolist()}, total {(end - start) / 100.0:.1f} us')
    for wave in (0, 7, 15):
        c = clk[w, wave]
        k = int(np.count_nonzero(c))
        if k == 0:
            print(f'   wave {wave:2d}: no stamps'); continue
        c = c[:k]
        d = np.diff(c) / 100.0
        print(f'   wave {wave:2d}: first stamp +{(c[0] - start) / 100.0:5.1f} us, {k - 1} iterations: ' + ' '.join(f'{x:.1f}' for x in d) + f' | after loop to end {(end - c[-1]) / 100.0:5.1f}')
