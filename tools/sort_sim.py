#!/usr/bin/env python3
"""Would ordering the lanes by their big-tile pattern cut k_td_play's L1 misses?

k_td_play is bound by L1 misses of its table gathers (TCP counters in profiles/): ~28 M 64-byte requests per launch.
This replays ONE CU's gather stream (4096 of 2^20 lanes, the cold tuples only — the hot ones come from LDS) through an
LRU model of the 32 KB L1 (256 lines of 128 B) for several lane orders and prints misses per lane."""
import collections
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module('2048_amd')
importlib.import_module('2048_amd.engine')
N = int(os.environ.get('N', 5))
F = pkg.engine.NUM_FEAT[N]
LANES = int(os.environ.get('LANES', 4096))


def slots_of(boards):
    e = pkg.Engine(len(boards), n=N, seed=1)
    e.set_boards(boards)
    f = e.features().astype(np.int64)
    offs, _ = pkg.engine.feature_layout(N)
    e.close()
    return f + offs[None, :]


def stream(eng, boards):
    after, _, changed = eng.boards_move_all(boards)
    s = slots_of(after.reshape(-1, 4, 4)).reshape(len(boards), 4, F)
    valid = ((changed[:, None] >> np.arange(4)[None, :]) & 1).astype(bool)
    return np.where(valid[:, :, None], s, -1)


def hot(slots):
    idx = slots & 0xFFFF
    q = (slots >= 0) & (slots < 17 * 65536)
    return q & ((idx & 0xF) <= 5) & (((idx >> 4) & 0xF) <= 5) & (((idx >> 8) & 0xF) <= 5) & (((idx >> 12) & 0xF) <= 5)


def l1_misses(s, lines=256, use_hot=True):
    """LRU over 128-byte lines; the CU's 12 resident waves advance together, instruction by instruction"""
    lru = collections.OrderedDict()
    miss = acc = 0
    for w0 in range(0, len(s), 64 * 12):
        blk = s[w0:w0 + 64 * 12]
        h = hot(blk) if use_hot else np.zeros(blk.shape, bool)
        for d in range(4):
            for f in range(F):
                for w in range(0, len(blk), 64):
                    a = blk[w:w + 64, d, f]
                    a = a[(a >= 0) & ~h[w:w + 64, d, f]]
                    for line in np.unique(a >> 5):
                        acc += 1
                        if line in lru:
                            lru.move_to_end(line)
                        else:
                            miss += 1
                            lru[line] = 1
                            if len(lru) > lines:
                                lru.popitem(last=False)
    return miss / len(s), acc / len(s)


def key_big(boards, thr):
    """lexicographic key on the tiles above `thr` (0 for the others), then on the whole board"""
    b = boards.reshape(len(boards), 16).astype(np.int64)
    big = np.where(b > thr, b, 0)
    k1 = np.zeros(len(b), np.int64)
    k2 = np.zeros(len(b), np.int64)
    for j in range(16):
        k1 = k1 * 16 + big[:, j]
        k2 = k2 * 16 + b[:, j]
    return np.lexsort((k2, k1))


def report(tag, eng, alpha):
    """sort by the pattern of tiles > 5 now, then keep that order for k more steps"""
    boards = eng.get_boards()
    B = len(boards)
    mid = B // 2
    base = np.arange(B)[mid:mid + LANES]
    m0, a0 = l1_misses(stream(eng, boards[base]))
    out = [f'{tag}: L1 misses per lane (accesses per lane), one CU = {LANES} lanes of {B}; unsorted {m0:.1f} ({a0:.1f})']
    order = key_big(boards, 5)
    sel = order[mid:mid + LANES]
    done = 0
    for k in (0, 2, 8, 32, 96):
        eng.td_steps(alpha, k - done)
        done = k
        b = eng.get_boards()
        m, a = l1_misses(stream(eng, b[sel]))
        fresh = key_big(b, 5)[mid:mid + LANES]
        m2, a2 = l1_misses(stream(eng, b[fresh]))
        out.append(f'  order {k:3d} steps old: {m:6.1f} ({a:6.1f})    freshly sorted: {m2:6.1f} ({a2:6.1f})')
    print('\n'.join(out), flush=True)


B = 1 << 20
eng = pkg.Engine(B, n=N, seed=2048)
eng.init_weights(seed=7, scale=0.01)
alpha = 0.25 * F / (8.0 * B)
eng.td_steps(alpha, 400)
report('fresh agent, 400 steps (bench window)', eng, alpha)
eng.set_update_rule(1)
eng.td_steps(0.25, 4000)
st = eng.stats()
report(f'mean rule +4000 steps (mean score {st["score_sum"] / max(1, st["episodes"]):.0f})', eng, 0.25)
eng.stats_reset()
eng.td_steps(0.25, 8000)
st = eng.stats()
report(f'mean rule +12000 steps (mean score {st["score_sum"] / max(1, st["episodes"]):.0f})', eng, 0.25)
