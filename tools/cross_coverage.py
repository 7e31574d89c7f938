#!/usr/bin/env python3
"""Which share of k_td_play's gathers (four-cell AND five-cell features, n = 5) would LDS-resident sets of several sizes
catch, and how many distinct 64-byte table lines a 64-lane wave asks for per gather instruction (what the L1 hands to L2 when
nothing hits).  Places are the table's memory order: t = 4 x 4 bit-transposed low 16 index bits; a cross adds its centre
cell on top (centre << 16 | t of up, left, down, right)."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module('2048_amd')
importlib.import_module('2048_amd.engine')
N = 5
F = pkg.engine.NUM_FEAT[N]


def transpose16(x):
    t = (x ^ (x >> 3)) & 0x0A0A
    x = x ^ t ^ (t << 3)
    t = (x ^ (x >> 6)) & 0x00CC
    return (x ^ t ^ (t << 6)) & 0xFFFF


def lines_per_wave(place, valid):
    """mean number of distinct 64-byte lines among the valid lanes of each 64-lane wave, per feature column"""
    L, D, K = place.shape
    ln = (place >> 4).reshape(L // 64, 64, D, K)
    va = valid.reshape(L // 64, 64, D)
    out = []
    for k in range(K):
        tot = 0
        for d in range(D):
            x = np.where(va[:, :, d], ln[:, :, d, k], -1)
            x = np.sort(x, axis=1)
            distinct = (np.diff(x, axis=1) != 0).sum(axis=1) + 1 - (x[:, 0] == -1)
            tot += distinct.sum()
        out.append(tot / max(1, va.sum()) * 64)
    return np.array(out)


def report(tag, eng, lanes=65536):
    boards = eng.get_boards()[:lanes]
    after, _, changed = eng.boards_move_all(boards)
    e = pkg.Engine(lanes * 4, n=N, seed=1)
    e.set_boards(after.reshape(-1, 4, 4))
    f = e.features().astype(np.int64).reshape(lanes, 4, F)
    e.close()
    valid = ((changed[:, None] >> np.arange(4)[None, :]) & 1).astype(bool)
    q = f[:, :, :17]                 # (Engine.features: every feature's index inside its own table)
    x = f[:, :, 17:]
    qt = transpose16(q)
    xt, xc = transpose16(x & 0xFFFF), x >> 16
    print(f'{tag}: {valid.sum()} valid directions of {lanes} lanes')
    qv, xtv, xcv = qt[valid], xt[valid], xc[valid]
    for lim in (256, 512, 1024, 2048, 4096):
        print(f'    four-cell  t < {lim:5d}: {lim * 17 * 4 // 1024:4d} KB  {np.mean(qv < lim):6.1%} of the four-cell gathers')
    for cl, tl in ((4, 256), (4, 1024), (6, 256), (8, 256), (8, 512), (8, 1024), (8, 4096), (16, 256), (16, 1024)):
        m = (xcv < cl) & (xtv < tl)
        print(f'    cross centre < {cl:2d}, t < {tl:5d}: {cl * tl * 16 // 1024:4d} KB  {m.mean():6.1%} of the five-cell gathers')
    lq = lines_per_wave(qt, valid)
    lx = lines_per_wave((xc << 16) | xt, valid)
    print(f'    distinct 64 B lines per full wave and gather: four-cell {lq.mean():.1f} (features {np.round(lq, 1).tolist()}), cross {lx.mean():.1f} ({np.round(lx, 1).tolist()})')
    cold = qt >= 2048
    lqc = lines_per_wave(np.where(cold, qt, 0), valid & True)      # rough: cold lanes only
    print(f'    lanes per wave that are cold (t >= 2048): {np.mean(cold[valid]) * 64:.1f} of 64')


B = 1 << 20
eng = pkg.Engine(B, n=N, seed=2048)
eng.init_weights(seed=7, scale=0.01)
alpha = 0.25 * F / (8.0 * B)
eng.td_steps(alpha, 320)
report('fresh agent (bench window), lanes sorted' if os.environ.get('G2048_SORT_EVERY', '') != '0' else 'fresh agent, unsorted', eng)
eng.set_update_rule(1)
eng.td_steps(0.25, 3000)
report('mean rule + 3000 steps', eng)
