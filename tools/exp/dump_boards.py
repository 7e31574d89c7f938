#!/usr/bin/env python3
"""Samples of the boards an n = 5 context holds (fresh agent in the bench window; mean rule + 3 000 and + 12 000 steps) for
offline studies of chunkings / orders: gpurun_out/board_samples.npz (131 072 boards each)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pkg = importlib.import_module('2048_amd')
N, B, K = 5, 1 << 20, 131072
F = 21
eng = pkg.Engine(B, n=N, seed=2048)
eng.init_weights(seed=7, scale=0.01)
eng.td_steps(0.25 * F / (8.0 * B), 320)
out = {'fresh': eng.get_boards()[:K].astype(np.uint8)}
eng.set_update_rule(1)
eng.td_steps(0.25, 3000)
out['mean3000'] = eng.get_boards()[:K].astype(np.uint8)
eng.td_steps(0.25, 9000)
out['mean12000'] = eng.get_boards()[:K].astype(np.uint8)
st = eng.stats()
print('mean score', st['score_sum'] / max(1, st['episodes']))
np.savez_compressed(os.path.join(ROOT, 'gpurun_out', 'board_samples.npz'), **out)
