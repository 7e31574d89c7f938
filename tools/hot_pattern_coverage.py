#!/usr/bin/env python3
"""Would a per-table choice of WHICH 2 048-entry block of a four-cell table sits in LDS beat the fixed first block?
k_td_play_hot keeps entries with t >> 11 == 0 (t = the entry's place in memory order: [bit 3 of the four cells | bit 2 | bit 1 |
bit 0]: all cells <= 128 and the first one <= 8).  For a trained agent: share of the four-cell gathers per value of t >> 11, per
table; coverage of block 0 against the best block of each table, and against the best TWO blocks (if LDS had room)."""
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


def report(tag, eng, lanes=131072):
    boards = eng.get_boards()[:lanes]
    after, _, changed = eng.boards_move_all(boards)
    e = pkg.Engine(lanes * 4, n=N, seed=1)
    e.set_boards(after.reshape(-1, 4, 4))
    f = e.features().astype(np.int64).reshape(lanes, 4, F)[:, :, :17]
    e.close()
    valid = ((changed[:, None] >> np.arange(4)[None, :]) & 1).astype(bool)
    t = transpose16(f[valid])                        # [gathers, 17]
    blk = t >> 11
    cov0, covb, cov2 = [], [], []
    for k in range(17):
        h = np.bincount(blk[:, k], minlength=32) / len(blk)
        order = np.argsort(-h)
        cov0.append(h[0]); covb.append(h[order[0]]); cov2.append(h[order[0]] + h[order[1]])
    print(f'{tag}: block 0 {np.mean(cov0):.3f} | best block per table {np.mean(covb):.3f} | best two blocks {np.mean(cov2):.3f} of the four-cell gathers')
    print('    per table, block 0 / best: ' + ' '.join(f'{a:.2f}/{b:.2f}' for a, b in zip(cov0, covb)))


B = 1 << 20
eng = pkg.Engine(B, n=N, seed=2048)
eng.init_weights(seed=7, scale=0.01)
eng.td_steps(0.25 * F / (8.0 * B), 320)
report('fresh agent (bench window)', eng)
eng.set_update_rule(1)
eng.td_steps(0.25, 3000)
report('mean rule + 3000 steps', eng)
eng.td_steps(0.25, 9000)
st = eng.stats()
report(f'mean rule + 12000 steps (mean score {st["score_sum"] / max(1, st["episodes"]):.0f})', eng)
