#!/usr/bin/env python3
"""k_td_play with the lanes physically ordered by their big-tile pattern (ordering done on the host here — this only
measures what the order is worth to the kernel, and how fast it wears off)."""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module('2048_amd')
importlib.import_module('2048_amd.engine')
N = int(os.environ.get('N', 5))
F = pkg.engine.NUM_FEAT[N]
THR = int(os.environ.get('THR', 5))


def key_big(boards, thr, secondary=True):
    b = boards.reshape(len(boards), 16).astype(np.int64)
    big = np.where(b > thr, b, 0)
    k1 = np.zeros(len(b), np.int64)
    k2 = np.zeros(len(b), np.int64)
    for j in range(16):
        k1 = k1 * 16 + big[:, j]
        k2 = k2 * 16 + b[:, j]
    if not secondary:
        return np.argsort(k1, kind='stable')           # what the device does: big-tile pattern only
    return np.lexsort((k2, k1))


def reorder(eng, order):
    b, s, r = eng.get_boards(), eng.get_scores(), eng.get_rng()
    eng.set_boards(b[order])            # clears the carry: the next step emits no records, its gathers are the same
    eng.set_scores(s[order])
    eng.set_rng(r[order])


def series(eng, alpha, tag):
    out = []
    for k in range(12):
        t = eng.td_steps_kernel_ms(alpha, 1)
        out.append(t[0] * 1e3)
    print(f'{tag}: k_td_play us per step after the reorder: ' + ' '.join(f'{x:5.0f}' for x in out), flush=True)


B = 1 << 20
eng = pkg.Engine(B, n=N, seed=2048)
eng.init_weights(seed=7, scale=0.01)
eng.set_lane_sort(0)
alpha = 0.25 * F / (8.0 * B)
eng.td_steps(alpha, 400)
series(eng, alpha, 'fresh agent, as is          ')
reorder(eng, np.arange(B))
series(eng, alpha, 'fresh agent, identity order ')
reorder(eng, key_big(eng.get_boards(), THR, secondary=False))
series(eng, alpha, 'fresh agent, pattern only   ')
reorder(eng, key_big(eng.get_boards(), THR))
series(eng, alpha, 'fresh agent, pattern + board')
eng.set_update_rule(1)
eng.td_steps(0.25, 6000)
st = eng.stats()
print(f'trained: mean score {st["score_sum"] / max(1, st["episodes"]):.0f}')
series(eng, 0.25, 'trained agent, as is        ')
reorder(eng, key_big(eng.get_boards(), THR, secondary=False))
series(eng, 0.25, 'trained agent, pattern only ')
reorder(eng, key_big(eng.get_boards(), THR))
series(eng, 0.25, 'trained agent, pattern+board')
reorder(eng, np.random.RandomState(1).permutation(B))
series(eng, 0.25, 'trained agent, shuffled     ')
