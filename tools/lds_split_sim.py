#!/usr/bin/env python3
"""How should the CU's LDS be split between the four-cell tables' hot set and a cross-table hot set?  Distinct 64-byte lines per
64-lane wave and step (what k_td_play's L1 misses follow) for every combination, lanes in the shipped order."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module('2048_amd')
importlib.import_module('2048_amd.engine')
N, F = 5, 21


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
    after, _, changed = eng.boards_move_all(boards[order])
    e = pkg.Engine(lanes * 4, n=N, seed=1)
    e.set_boards(after.reshape(-1, 4, 4))
    f = e.features().astype(np.int64).reshape(lanes, 4, F)
    e.close()
    valid = ((changed[:, None] >> np.arange(4)[None, :]) & 1).astype(bool)
    q = transpose16(f[:, :, :17])               # (Engine.features: every feature's index inside its own table)
    x = f[:, :, 17:]
    xc, xt = x >> 16, transpose16(x & 0xFFFF)
    xline = (xc << 12) | (xt >> 4)
    cells = np.stack([(x >> s) & 15 for s in (0, 4, 8, 12, 16)], axis=-1)
    waves = lanes // 64
    qres, xres = {}, {}
    for hot in (512, 1024, 1536, 2048):
        qres[hot] = sum(distinct_per_wave(q[:, d, :] >> 4, valid[:, d, None] & (q[:, d, :] >= hot)) for d in range(4)) / waves
    rules = {'none (0 KB)': np.zeros_like(xc, bool), 'centre < 4, t < 256 (16 KB)': (xc < 4) & (xt < 256), 'centre < 8, t < 256 (32 KB)': (xc < 8) & (xt < 256),
             'any centre, t < 256 (64 KB)': xt < 256, 'centre < 4, t < 1024 (64 KB)': (xc < 4) & (xt < 1024),
             'all five cells <= 4 (50 KB)': (cells <= 4).all(axis=-1), 'all five cells <= 5 (124 KB)': (cells <= 5).all(axis=-1),
             'centre < 8, t < 1024 (128 KB)': (xc < 8) & (xt < 1024)}
    for name, hotm in rules.items():
        xres[name] = (sum(distinct_per_wave(xline[:, d, :], valid[:, d, None] & ~hotm[:, d, :]) for d in range(4)) / waves, hotm[valid].mean())
    print('   four-cell hot entries per table -> cold lines per wave and step:', {h: round(v, 1) for h, v in qres.items()})
    for name, (v, cov) in xres.items():
        print(f'   cross set {name:34s} covers {cov:6.1%} of the cross gathers; cold cross lines {v:7.1f};  total with four-cell 512 / 1024 / 1536 / 2048: ' +
              ' / '.join(f'{v + qres[h]:7.1f}' for h in (512, 1024, 1536, 2048)), flush=True)


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
