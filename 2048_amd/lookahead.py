"""Batched sampled expectimax with a CALLER-SUPPLIED sampler — `Game.look_forward` (game2048/game_logic.py:214-243) for many
positions at once.

The product path is on the device: `Engine.boards_look_forward` / `Engine.lookahead_steps` (csrc/lookahead.hip) expand, evaluate
and reduce every tree in HBM and draw the chance nodes from the device's own spec (rng.lookahead_draws); `Game._find_best_move`
and `QAgent.trial(depth > 0)` use those.  This module is the form for a sampler the device cannot know (a test that injects
draws keyed some other way): a whole level of the tree is one batch, the children of every node are built on the host, all
their moves come from one `g2048_boards_move_all` call and all leaf values from one `g2048_boards_evaluate` call.

    V_0(s)            = estimator(s)
    V_d(s), d > 0     = estimator(s)                                  if empty(s) >= since_empty
                      = mean over min(width, empty(s)) sampled tiles t of max(0, W(s + t))
    W(c)              = -100                                          if c is game over
                      = max over directions that change c of V_{d-1}(after_dir(c))

Sampling (which empty cells, 90/10 tile) is delegated to a `sampler(rows, k) -> (cells, tiles)`; the default draws from
a NumPy generator.  The reference draws in depth-first order from Python's `random`, so individual samples differ from
it; the value recursion above is the reference's, and is what the tests pin (with a sampler keyed by the board).
"""
import numpy as np


def random_sampler(rng=None):
    rng = rng or np.random.default_rng()

    def sample(rows, k):
        """rows [M,4,4]; k [M] tiles wanted per row (<= its empty cells).  Returns cells [M,kmax] (flat index, -1 =
        unused) and tiles [M,kmax]."""
        M, kmax = len(rows), int(k.max()) if len(k) else 0
        flat = rows.reshape(M, 16)
        keys = rng.random((M, 16))
        keys[flat != 0] = 2.0                                   # occupied cells sort last
        order = np.argsort(keys, axis=1)[:, :kmax]              # kmax distinct empty cells per row, uniformly
        cells = np.where(np.arange(kmax)[None, :] < k[:, None], order, -1)
        tiles = np.where(rng.random((M, kmax)) < 0.1, 2, 1)     # 1 if random.randrange(10) else 2 (game_logic.py:225)
        return cells, tiles
    return sample


def expectimax_values(engine, rows, depth, width, since_empty, sampler):
    """V_depth of every board in rows [N,4,4] (uint8) with the weight table of `engine`."""
    rows = np.ascontiguousarray(rows, dtype=np.uint8).reshape(-1, 4, 4)
    N = len(rows)
    if N == 0:
        return np.zeros(0, np.float64)
    n_empty = 16 - np.count_nonzero(rows.reshape(N, 16), axis=1)
    leaf = (n_empty >= since_empty) if depth > 0 else np.ones(N, bool)
    values = np.zeros(N, np.float64)
    if leaf.any():
        values[leaf] = engine.boards_evaluate(rows[leaf])
    inner = np.nonzero(~leaf)[0]
    if len(inner) == 0:
        return values
    k = np.minimum(width, n_empty[inner])
    if (k == 0).any():
        raise ZeroDivisionError('look_forward on a full board (the reference divides by zero here, game_logic.py:242)')
    cells, tiles = sampler(rows[inner], k)
    par, slot = np.nonzero(cells >= 0)                          # one child per (inner node, sampled cell)
    child = rows[inner][par].reshape(-1, 16).copy()
    child[np.arange(len(child)), cells[par, slot]] = tiles[par, slot]
    after, _, changed = engine.boards_move_all(child)
    W = np.full(len(child), -100.0)                             # dead position (game over after the new tile)
    ci, di = np.nonzero((changed[:, None] >> np.arange(4)[None, :]) & 1)
    if len(ci):
        sub = expectimax_values(engine, after[ci, di], depth - 1, width, since_empty, sampler)
        best = np.full(len(child), -np.inf)
        np.maximum.at(best, ci, sub)
        alive = np.isfinite(best)
        W[alive] = best[alive]
    total = np.zeros(len(inner))
    np.add.at(total, par, np.maximum(W, 0.0))
    values[inner] = total / k
    return values
