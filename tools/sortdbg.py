import importlib, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module('2048_amd')
n, B, steps = 4, 1 << 17, 40
out = []
for every in (0, 0, 3, 3):
    eng = pkg.Engine(B, n=n, seed=12)
    eng.init_weights(seed=3, scale=0.01)
    eng.set_lane_sort(every)
    eng.td_steps(0.25 * 17 / (8 * B), steps)
    out.append((every, eng.get_boards().reshape(B, 16), eng.get_weights()))
    eng.close()
for i in range(4):
    for j in range(i + 1, 4):
        d = (out[i][1] != out[j][1]).any(axis=1).sum()
        print('every', out[i][0], 'vs', out[j][0], ': lanes with different boards', d, ' max |dw|', np.abs(out[i][2] - out[j][2]).max())
