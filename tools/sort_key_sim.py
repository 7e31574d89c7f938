#!/usr/bin/env python3
"""How many distinct 64-byte table lines does a 64-lane wave of k_td_play ask for per gather instruction, under different lane
orders?  (The L1's tag look-ups and misses, which bound the kernel, follow this number.)  Boards of the bench workload; every
candidate order is a sort of the lanes by a key computed from the board; a wave is 64 consecutive lanes of the order."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module('2048_amd')
importlib.import_module('2048_amd.engine')
N = 5
F = pkg.engine.NUM_FEAT[N]
HOT = 2048


def transpose16(x):
    t = (x ^ (x >> 3)) & 0x0A0A
    x = x ^ t ^ (t << 3)
    t = (x ^ (x >> 6)) & 0x00CC
    return (x ^ t ^ (t << 6)) & 0xFFFF


def distinct_per_wave(lines, active):
    """lines, active: [L, K]; returns the total number of distinct lines among the active lanes of each 64-lane wave, summed over K"""
    L, K = lines.shape
    x = np.where(active, lines, -1).reshape(L // 64, 64, K)
    x = np.sort(x, axis=1)
    d = (np.diff(x, axis=1) != 0).sum(axis=1) + 1 - (x[:, 0, :] == -1)
    return d.sum()


def mix16(k):
    """a 16-bit hash of a 64-bit key (a stand-in for the kernel's multiplicative hash: what matters is that equal keys meet)"""
    k = np.asarray(k, np.uint64)
    k = (k ^ (k >> np.uint64(31))) * np.uint64(0x9E3779B97F4A7C15)
    k = (k ^ (k >> np.uint64(29))) * np.uint64(0xBF58476D1CE4E5B9)
    return ((k ^ (k >> np.uint64(32))) & np.uint64(0xFFFF)).astype(np.int64)


def lex(h):
    k = np.zeros(len(h), np.int64)
    for c in range(16):
        k = k * 8 + h[:, c]
    return k


def orders(boards):
    """name -> permutation of the lanes"""
    b = boards.reshape(len(boards), 16).astype(np.int64)
    shuffled = np.random.RandomState(1).permutation(len(b))         # the counting sort leaves the order inside a bucket arbitrary
    by = lambda key: shuffled[np.argsort(key[shuffled], kind='stable')]
    out = {'lane id (no sort)': np.arange(len(b))}
    for thr in (5, 4, 6):
        big = b > thr
        pos = (big << np.arange(15, -1, -1)).sum(axis=1)
        full = lex(np.where(big, b >> 1, 0))
        if thr == 5:
            out['one bit per cell, tile > 32 (round 2)'] = by(pos)
            out['value >> 1 of tiles > 32, lexicographic, global sort'] = np.argsort(full, kind='stable')
            out['the same string hashed to 16 bits (shipped)'] = by(mix16(full))
            o = by(pos).copy()
            for lo in range(0, len(o), 4096):
                seg = o[lo:lo + 4096]
                o[lo:lo + 4096] = seg[np.lexsort((full[seg], pos[seg]))]
            out['round-2 key, then every 4096-lane tile sorted by the string'] = o
            out['value (not >> 1) of tiles > 32, hashed'] = by(mix16(lex(np.where(big, b, 0) & 7) * 16 + (lex(np.where(big, b >> 3, 0)) & 0xFFFF)))
        else:
            out[f'value >> 1 of tiles > {1 << thr}, hashed'] = by(mix16(full))
    return out


def report(tag, eng, lanes=1 << 20):
    boards = eng.get_boards()[:lanes]
    for name, order in orders(boards).items():
        bs = boards[order]
        after, _, changed = eng.boards_move_all(bs)
        e = pkg.Engine(lanes * 4, n=N, seed=1)
        e.set_boards(after.reshape(-1, 4, 4))
        f = e.features().astype(np.int64).reshape(lanes, 4, F)
        e.close()
        valid = ((changed[:, None] >> np.arange(4)[None, :]) & 1).astype(bool)
        q = transpose16(f[:, :, :17])           # (Engine.features: every feature's index inside its own table)
        x = f[:, :, 17:]
        xp = ((x >> 16) << 16) | transpose16(x & 0xFFFF)
        tq = tx = 0
        for d in range(4):
            tq += distinct_per_wave(q[:, d, :] >> 4, valid[:, d, None] & (q[:, d, :] >= HOT))
            tx += distinct_per_wave(xp[:, d, :] >> 4, np.repeat(valid[:, d, None], 4, axis=1))
        waves = lanes // 64
        print(f'  {name:58s} lines per wave and step: cold four-cell {tq / waves:7.1f}  cross {tx / waves:7.1f}  total {(tq + tx) / waves:7.1f}', flush=True)


B = 1 << 20
eng = pkg.Engine(B, n=N, seed=2048)
eng.init_weights(seed=7, scale=0.01)
alpha = 0.25 * F / (8.0 * B)
eng.td_steps(alpha, 320)
print('fresh agent (bench window)')
report('fresh', eng)
eng.set_update_rule(1)
eng.td_steps(0.25, 3000)
print('mean rule + 3000 steps')
report('trained', eng)
