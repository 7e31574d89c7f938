#!/usr/bin/env python3
"""Which 16 entries of a four-cell table should share a 64-byte line?  Distinct lines per 64-lane wave and step for the cold
four-cell gathers of k_td_play (cold as shipped: place >= 2048 in table_place order) under candidate groupings, lanes in the
shipped order (hash of value >> 1 of the tiles above 32).  The index is a << 12 | b << 8 | c << 4 | d."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module('2048_amd')
importlib.import_module('2048_amd.engine')
sim = importlib.import_module('tools.sort_key_sim') if False else None
N, F, HOT = 5, 21, 2048


def transpose16(x):
    t = (x ^ (x >> 3)) & 0x0A0A
    x = x ^ t ^ (t << 3)
    t = (x ^ (x >> 6)) & 0x00CC
    return (x ^ t ^ (t << 6)) & 0xFFFF


def distinct_per_wave(lines, active):
    L, K = lines.shape
    x = np.where(active, lines, -1).reshape(L // 64, 64, K)
    x = np.sort(x, axis=1)
    return ((np.diff(x, axis=1) != 0).sum(axis=1) + 1 - (x[:, 0, :] == -1)).sum()


def groupings(idx):
    a, b, c, d = (idx >> 12) & 15, (idx >> 8) & 15, (idx >> 4) & 15, idx & 15
    g = {}
    g['shipped: bit 0 of a, b, c, d in the line'] = (a >> 1) << 9 | (b >> 1) << 6 | (c >> 1) << 3 | (d >> 1)
    g['low 2 bits of c and d'] = a << 8 | b << 4 | (c >> 2) << 2 | (d >> 2)
    g['low 2 bits of a and b'] = (a >> 2) << 10 | (b >> 2) << 8 | c << 4 | d
    g['low 2 bits of b and c'] = a << 8 | (b >> 2) << 6 | (c >> 2) << 4 | d
    g['all 4 bits of d (index order)'] = a << 8 | b << 4 | c
    g['all 4 bits of a'] = b << 8 | c << 4 | d
    g['low 2 of d, bit 0 of b and c'] = a << 8 | (b >> 1) << 5 | (c >> 1) << 2 | (d >> 2)
    mn = np.minimum(np.minimum(a, b), np.minimum(c, d))
    return g


def mix16(k):
    k = np.asarray(k, np.uint64)
    k = (k ^ (k >> np.uint64(31))) * np.uint64(0x9E3779B97F4A7C15)
    k = (k ^ (k >> np.uint64(29))) * np.uint64(0xBF58476D1CE4E5B9)
    return ((k ^ (k >> np.uint64(32))) & np.uint64(0xFFFF)).astype(np.int64)


def report(eng, lanes=1 << 19):
    boards = eng.get_boards()[:lanes]
    b = boards.reshape(lanes, 16).astype(np.int64)
    k = np.zeros(lanes, np.int64)
    for c in range(16):
        k = k * 8 + np.where(b[:, c] > 5, b[:, c] >> 1, 0)
    order = np.argsort(mix16(k), kind='stable')
    bs = boards[order]
    after, _, changed = eng.boards_move_all(bs)
    e = pkg.Engine(lanes * 4, n=N, seed=1)
    e.set_boards(after.reshape(-1, 4, 4))
    f = e.features().astype(np.int64).reshape(lanes, 4, F)
    e.close()
    valid = ((changed[:, None] >> np.arange(4)[None, :]) & 1).astype(bool)
    q = f[:, :, :17]                # (Engine.features: every feature's index inside its own table)
    cold = transpose16(q) >= HOT
    for name, line in groupings(q).items():
        tot = 0
        for d in range(4):
            tot += distinct_per_wave(line[:, d, :], valid[:, d, None] & cold[:, d, :])
        print(f'  {name:44s} cold four-cell lines per wave and step {tot / (lanes // 64):7.1f}', flush=True)


B = 1 << 20
eng = pkg.Engine(B, n=N, seed=2048)
eng.init_weights(seed=7, scale=0.01)
eng.td_steps(0.25 * F / (8.0 * B), 320)
print('fresh agent (bench window)')
report(eng)
eng.set_update_rule(1)
eng.td_steps(0.25, 3000)
print('mean rule + 3000 steps')
report(eng)
