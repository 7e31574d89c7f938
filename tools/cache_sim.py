#!/usr/bin/env python3
"""How much of k_td_play's table-gather stream would a per-CU software cache in LDS absorb?

One CU plays ~4096 lanes per launch (2^20 lanes / 256 CUs): 4 directions x num_feat gathers each.  This replays that
stream (wave by wave, 64 lanes per instruction, feature-major as the kernel issues them) through a direct-mapped
{tag, value} cache of E entries and reports the hit rate, for a fresh agent (the bench's state) and after training.
Boards, afterstates and slots come from the device through the C ABI."""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module('2048_amd')
importlib.import_module('2048_amd.engine')
N = int(os.environ.get('N', 5))
F = pkg.engine.NUM_FEAT[N]


def slots_of(boards):
    """flat table slots [count, F] of any boards, via g2048_features on a scratch engine"""
    e = pkg.Engine(len(boards), n=N, seed=1)
    e.set_boards(boards)
    f = e.features().astype(np.int64)
    offs, _ = pkg.engine.feature_layout(N)
    e.close()
    return f + offs[None, :]


def stream(eng, lanes):
    b = eng.get_boards()[:lanes]
    after, _, changed = eng.boards_move_all(b)
    s = slots_of(after.reshape(-1, 4, 4)).reshape(lanes, 4, F)
    valid = ((changed[:, None] >> np.arange(4)[None, :]) & 1).astype(bool)
    s = np.where(valid[:, :, None], s, 0)                    # invalid directions read slot 0, as the kernel does
    return s


def simulate(s, entries, waves_in_flight=12):
    """direct-mapped cache; the CU's 12 resident waves progress together: instruction (d, f) of all 12 waves, then the next"""
    lanes = s.shape[0]
    tags = np.full(entries, -1, np.int64)
    hits = total = 0
    lines = 0
    mask = entries - 1
    for w0 in range(0, lanes, 64 * waves_in_flight):
        blk = s[w0:w0 + 64 * waves_in_flight]
        for d in range(4):
            for f in range(F):
                for w in range(0, len(blk), 64):
                    a = blk[w:w + 64, d, f]
                    h = (a ^ (a >> 14) ^ (a >> 7)) & mask
                    hit = tags[h] == a
                    hits += int(hit.sum())
                    total += len(a)
                    miss = a[~hit]
                    lines += len(np.unique(miss >> 5))       # distinct 128-byte lines the miss lanes still need
                    tags[h[~hit]] = miss
    return hits / total, lines / (total / 64)


def report(tag, eng):
    s = stream(eng, 4096)
    uniq = len(np.unique(s)) / s.size
    full_lines = np.mean([len(np.unique(s[w:w + 64, d, f] >> 5)) for w in range(0, 4096, 64) for d in range(4) for f in range(F)])
    out = [f'{tag}: distinct slots {uniq:.3f} of accesses, lines per wave instruction now {full_lines:.1f}']
    for e in (4096, 8192, 16384):
        hr, lines = simulate(s, e)
        out.append(f'  cache {e:6d} entries: hit rate {hr:.3f}, lines per wave instruction left {lines:.1f}')
    quad = s[:, :, :17]
    out.append(f'  fixed hot set (four-cell tuples with all tiles <= 5): {np.mean(hot(quad)):.3f} of the four-cell reads')
    print('\n'.join(out), flush=True)


def hot(slots):
    idx = slots & 0xFFFF
    return ((idx & 0xF) <= 5) & (((idx >> 4) & 0xF) <= 5) & (((idx >> 8) & 0xF) <= 5) & (((idx >> 12) & 0xF) <= 5)


B = 1 << 18
eng = pkg.Engine(B, n=N, seed=2048)
eng.init_weights(seed=7, scale=0.01)
alpha = 0.25 * F / (8.0 * B)
eng.td_steps(alpha, 320)
report('fresh agent, 320 steps (bench window)', eng)
eng.td_steps(alpha, 700)
report('fresh agent, 1020 steps', eng)
eng.set_update_rule(1)
for k in range(3):
    eng.td_steps(0.25, 3000)
    st = eng.stats()
    report(f'mean rule +3000 steps (mean score {st["score_sum"] / max(1, st["episodes"]):.0f})', eng)
    eng.stats_reset()
