#!/usr/bin/env python3
"""Which share of k_td_play's four-cell gathers would an LDS-resident hot set catch?  Candidate sets, in the table's memory
order (4 x 4 bit-transposed index t = [bit 3 of the cells | bit 2 | bit 1 | bit 0]): t < 256 (every cell <= 3: empty, 2, 4, 8),
t < 1024, t < 4096 (every cell <= 7), and the round-2 set "every cell <= 5" (1 296 entries, needs a base-6 index)."""
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


def report(tag, eng, lanes=65536):
    boards = eng.get_boards()[:lanes]
    after, _, changed = eng.boards_move_all(boards)
    e = pkg.Engine(lanes * 4, n=N, seed=1)
    e.set_boards(after.reshape(-1, 4, 4))
    f = e.features().astype(np.int64).reshape(lanes, 4, F)[:, :, :17]
    e.close()
    valid = ((changed[:, None] >> np.arange(4)[None, :]) & 1).astype(bool)
    idx = f[valid]                                   # [gathers, 17] logical four-cell indices
    t = transpose16(idx)
    cells = np.stack([(idx >> s) & 15 for s in (0, 4, 8, 12)], axis=-1)
    tot = idx.size
    share = 17 / 21
    print(f'{tag}: {tot} four-cell gathers of valid directions ({share:.0%} of all gathers)')
    for name, m in (('t < 256  (cells <= 3), 17 KB', t < 256), ('t < 1024, 70 KB', t < 1024), ('t < 4096 (cells <= 7), 278 KB', t < 4096),
                    ('cells <= 5 (1 296 entries), 88 KB', (cells <= 5).all(axis=-1))):
        print(f'    {name:36s} {m.mean():6.1%} of the four-cell gathers = {m.mean() * share:6.1%} of all')


B = 1 << 20
eng = pkg.Engine(B, n=N, seed=2048)
eng.init_weights(seed=7, scale=0.01)
alpha = 0.25 * F / (8.0 * B)
eng.td_steps(alpha, 320)
report('fresh agent (bench window)', eng)
eng.set_update_rule(1)
eng.td_steps(0.25, 3000)
report('mean rule + 3000 steps', eng)
eng.td_steps(0.25, 6000)
st = eng.stats()
report(f'mean rule + 9000 steps (mean score {st["score_sum"] / max(1, st["episodes"]):.0f})', eng)
