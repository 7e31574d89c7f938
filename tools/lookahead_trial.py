#!/usr/bin/env python3
"""The reference's best published result (README.md:131-146: a 5-tuple agent + expectimax(depth 3, width 4, since_empty 6), 100
games, "1 second per move") on the device: train an agent in this run (mean rule), then QAgent.trial greedy and with the
look-ahead — every game's tree expanded, evaluated and reduced in HBM (csrc/lookahead.hip) — and print the README's table with
ms per move beside it.

    python tools/lookahead_trial.py [n] [batch] [episodes] [games] [depth] [width] [since_empty]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from game2048.r_learning import *   # noqa
arg = lambda i, d: type(d)(sys.argv[i]) if len(sys.argv) > i else d
n, batch, episodes, games = arg(1, 5), arg(2, 1 << 18), arg(3, 10_000_000), arg(4, 100)
depth, width, since_empty = arg(5, 3), arg(6, 4), arg(7, 6)
agent = QAgent(name=f'a{n}', storage='local', console='local', n=n, alpha=0.25, batch=batch, seed=1, decay_step=episodes // 4, rule='mean')
agent.print = lambda *a, **k: None
t0 = time.time()
agent.train_run(num_eps=episodes, saving=False)
print(f'n = {n}: trained {agent.step} episodes on {batch} lanes in {time.time() - t0:.1f} s (mean rule)', flush=True)


def table(title, results, seconds):
    tiles = np.array([1 << int(np.max(g.row)) for g in results])
    moves = sum(g.odometer for g in results)
    shares = ' '.join(f'{lim}: {(tiles >= lim).mean() * 100:.1f} %' for lim in (1024, 2048, 4096, 8192, 16384))
    print(f'{title}: {len(results)} games, average score {np.mean([g.score for g in results]):.0f}, best {max(g.score for g in results)}; {shares}', flush=True)
    print(f'    {moves} moves in {seconds:.2f} s = {seconds / moves * 1e3:.4f} ms per move ({moves / len(results):.0f} moves per game; all games in lock step)', flush=True)


quiet = dict(storage='local', console='web', log_file='trial_log.txt')
import builtins
real_print = builtins.print
for title, kw in ((f'greedy (depth 0)', dict()), (f'expectimax(depth {depth}, width {width}, since_empty {since_empty})', dict(depth=depth, width=width, since_empty=since_empty))):
    builtins.print = lambda *a, **k: None
    t0 = time.time()
    try:
        res = QAgent.trial(estimator=agent.evaluate, num=games, storage='local', console='local', **kw)
    finally:
        builtins.print = real_print
    table(title, res, time.time() - t0)
    best = res[0]
    best.moves.append(-1)
    chain = best.replay(verbose=False)                          # the record of the best game replays to its end
    assert np.array_equal(chain[best.odometer][0], best.row) and chain[best.odometer][1] == best.score
print('reference (README.md:131-146): 5-tuple agent after 100 k episodes + expectimax(3, 4, 6): average 69 743, 2048 96 %, 4096 79 %, 8192 18 %, "1 second per move"')
