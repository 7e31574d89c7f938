#!/usr/bin/env python3
"""How many distinct 128-byte lines does one wave-wide table gather of k_td_play touch, and how many of them miss a 32 KB
LRU L1, under different orders of the 16 index bits of a four-cell table entry?

The L1 looks up one tag per distinct line and instruction (TCP_TOTAL_CACHE_ACCESSES = 57 per gather), so the number of
distinct lines per instruction is what the kernel pays for.  A line holds 32 consecutive slots = the low 5 index bits:
today those are cell 3's four bits + one bit of cell 2, and the high bits of a cell hardly vary (tiles above 128 are rare).
Layouts tried here put the LOW bits of several cells into the line offset instead."""
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


def bitperm(order):
    """order[k] = which source bit lands at bit k of the new 16-bit index"""
    def f(s):
        lo = s & 0xFFFF
        out = np.zeros_like(lo)
        for k, src in enumerate(order):
            out |= ((lo >> src) & 1) << k
        return np.where(s >= 0, (s & ~0xFFFF) | out, s)
    return f


# cell j occupies source bits 4j .. 4j+3 (cell 0 = lowest nibble)
LAYOUTS = {
    'as is (cell 0 whole + 1 bit of cell 1 in the line)': bitperm(list(range(16))),
    'low 2 bits of each cell first, high 2 bits behind': bitperm([0, 1, 4, 5, 8, 9, 12, 13, 2, 3, 6, 7, 10, 11, 14, 15]),
    'low 3 bits of each cell first, bit 3 of each behind': bitperm([0, 1, 2, 4, 5, 6, 8, 9, 10, 12, 13, 14, 3, 7, 11, 15]),
    'bit 0 of the 4 cells, bit 1 of the 4 cells, bit 2 .., bit 3 ..': bitperm([0, 4, 8, 12, 1, 5, 9, 13, 2, 6, 10, 14, 3, 7, 11, 15]),
    'low 2 bits of cells 0,1 + bit 0 of cell 2 first': bitperm([0, 1, 4, 5, 8, 2, 3, 6, 7, 9, 10, 11, 12, 13, 14, 15]),
}


def measure(s, lines=256):
    """distinct lines per wave instruction, and misses of an LRU L1 shared by a CU's 8 resident waves"""
    lru = collections.OrderedDict()
    look = miss = inst = 0
    for w0 in range(0, len(s), 64 * 8):
        blk = s[w0:w0 + 64 * 8]
        for d in range(4):
            for f in range(F):
                for w in range(0, len(blk), 64):
                    a = blk[w:w + 64, d, f]
                    a = a[a >= 0]
                    if not len(a):
                        continue
                    inst += 1
                    for line in np.unique(a >> 5):
                        look += 1
                        if line in lru:
                            lru.move_to_end(line)
                        else:
                            miss += 1
                            lru[line] = 1
                            if len(lru) > lines:
                                lru.popitem(last=False)
    return look / inst, miss / inst


def key_big(boards, thr=5):
    b = boards.reshape(len(boards), 16).astype(np.int64)
    big = np.where(b > thr, b, 0)
    k1 = np.zeros(len(b), np.int64)
    for j in range(16):
        k1 = k1 * 16 + big[:, j]
    return np.argsort(k1, kind='stable')


def report(tag, eng):
    boards = eng.get_boards()
    B = len(boards)
    mid = B // 2
    for oname, sel in (('lanes as they are', np.arange(B)[mid:mid + LANES]), ('lanes ordered by big-tile pattern', key_big(boards)[mid:mid + LANES])):
        s = stream(eng, boards[sel])
        print(f'{tag}; {oname}: distinct lines / L1 misses per wave gather', flush=True)
        for name, fn in LAYOUTS.items():
            l, m = measure(fn(s))
            print(f'    {name:66s} {l:5.1f} / {m:5.1f}', flush=True)


B = 1 << 20
eng = pkg.Engine(B, n=N, seed=2048)
eng.init_weights(seed=7, scale=0.01)
eng.set_lane_sort(0)
alpha = 0.25 * F / (8.0 * B)
eng.td_steps(alpha, 400)
report('fresh agent, 400 steps (bench window)', eng)
eng.set_update_rule(1)
eng.td_steps(0.25, 6000)
st = eng.stats()
report(f'mean rule +6000 steps (mean score {st["score_sum"] / max(1, st["episodes"]):.0f})', eng)
