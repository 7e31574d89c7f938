#!/usr/bin/env python3
"""How much of k_td_play's four-cell gathers COULD an LDS hot set of K entries per table take?  Plays the bench workload (fresh
agent, sum rule) and a trained agent (mean rule), takes all 2^20 boards, forms the afterstates of every direction that moves and
counts, per four-cell table, how the gathers spread over its 65 536 entries: coverage of the shipped rule (the first 2 048
entries in table_place order), of the K most frequent entries (the ceiling for any K-entry hot set), and the Good-Turing
estimate of the mass the sample has not seen."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module('2048_amd')
B = int(os.environ.get('B', 1 << 20))


def feats():
    f = []
    for c in range(4): f.append([(0, c), (1, c), (2, c), (3, c)])
    for r in range(4): f.append([(r, 0), (r, 1), (r, 2), (r, 3)])
    for r in range(3):
        for c in range(3): f.append([(r, c), (r + 1, c), (r, c + 1), (r + 1, c + 1)])
    return f


def report(eng, tag):
    boards = eng.get_boards().reshape(B, 16)
    after, _, changed = eng.boards_move_all(boards)
    ci, di = np.nonzero((changed[:, None] >> np.arange(4)[None, :]) & 1)
    a = after[ci, di].reshape(-1, 4, 4)
    M = len(a)
    rows, fam = [], []
    for cells in feats():
        v = [a[:, r, c].astype(np.int64) for r, c in cells]
        key = (v[0] << 12) | (v[1] << 8) | (v[2] << 4) | v[3]
        cnt = np.bincount(key, minlength=65536)
        top = np.sort(cnt)[::-1]
        shipped = ((v[0] < 4) & (v[1] < 8) & (v[2] < 8) & (v[3] < 8)).mean()      # t < 2048 in [b3 | b2 | b1 | b0] order
        rows.append([shipped] + [top[:k].sum() / M for k in (512, 1024, 2048, 4096)] + [(cnt == 1).sum() / M, (cnt > 0).sum()])
        vv = np.stack(v, axis=-1)
        big = (vv >= 8).sum(axis=1)
        srt = np.sort(vv, axis=1)
        fam.append([(srt[:, 3] < 4).mean(), (srt[:, 3] < 6).mean(), (srt[:, 3] < 8).mean(),
                    ((big == 1) & (srt[:, 2] < 4)).mean(), ((big == 1) & (srt[:, 2] < 6)).mean(), ((big == 1) & (srt[:, 2] < 8)).mean(),
                    ((big == 2) & (srt[:, 1] < 4)).mean(), (big >= 2).mean(),
                    ((srt[:, 3] >= 6) & (srt[:, 2] < 4)).mean(), ((srt[:, 3] >= 6) & (srt[:, 2] < 6)).mean()])
    r = np.array(rows)
    print(f'{tag}: {M} gathers per table; max tile {boards.max()}; mean over the 17 tables:')
    print(f'   shipped rule (first 2048 in memory order) {r[:, 0].mean():.3f}   most frequent 512 / 1024 / 2048 / 4096 entries: '
          f'{r[:, 1].mean():.3f} / {r[:, 2].mean():.3f} / {r[:, 3].mean():.3f} / {r[:, 4].mean():.3f}   unseen mass (Good-Turing) {r[:, 5].mean():.4f}   '
          f'distinct entries used {int(r[:, 6].mean())}', flush=True)
    f = np.array(fam).mean(axis=0)
    print('   structural families (share of the gathers): all<4 %.3f  all<6 %.3f  all<8 %.3f | exactly one cell >= 8 and the others <4 %.3f  <6 %.3f  <8 %.3f | two cells >= 8, others <4 %.3f | two or more >= 8: %.3f | largest >= 6, others <4: %.3f  <6: %.3f' % tuple(f), flush=True)


eng = pkg.Engine(B, n=5, seed=2048)
eng.init_weights(seed=7, scale=0.01)
eng.td_steps(0.25 * 21 / (8.0 * B), 320)
report(eng, 'fresh agent, sum rule, 320 steps (the bench workload)')
eng.set_update_rule(1)
for steps in (1000, 3000, 8000):
    eng.td_steps(0.25, steps)
    st = eng.stats()
    report(eng, f'mean rule, +{steps} steps (mean score so far {st["score_sum"] / max(1, st["episodes"]):.0f})')
    eng.stats_reset()
