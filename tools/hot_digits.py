#!/usr/bin/env python3
"""How concentrated are the table reads?  Fraction of the 17 four-cell tuples (rows, columns, 2x2 squares) of the
boards in flight whose tiles are all <= t, for a fresh agent (the bench's state) and after some training."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module('2048_amd')


def quads(b):                       # b [B,4,4] -> [B,17] max tile of each four-cell tuple
    rows = b.max(axis=2)
    cols = b.max(axis=1)
    sq = np.stack([b[:, i:i + 2, j:j + 2].reshape(len(b), 4).max(axis=1) for i in range(3) for j in range(3)], axis=1)
    return np.concatenate([rows, cols, sq], axis=1)


def report(tag, eng):
    b = eng.get_boards().reshape(-1, 4, 4)
    q = quads(b)
    print(tag, 'mean max tile %.2f' % b.reshape(len(b), 16).max(axis=1).mean(),
          ' '.join('<=%d: %.3f' % (t, (q <= t).mean()) for t in (3, 4, 5, 6, 7, 9, 11)), flush=True)


B = 1 << 16
eng = pkg.Engine(B, n=5, seed=2048)
eng.init_weights(seed=7, scale=0.01)
alpha = 0.25 * 21 / (8.0 * B)
for steps in (64, 200, 400):
    eng.td_steps(alpha, steps)
    report('fresh sum-rule after +%d steps' % steps, eng)
eng.set_update_rule(1)
for k in range(6):
    eng.td_steps(0.25, 1500)
    st = eng.stats()
    report('mean-rule +1500 steps (episodes %d, mean score %.0f)' % (st['episodes'], st['score_sum'] / max(1, st['episodes'])), eng)
    eng.stats_reset()
