#!/usr/bin/env python3
"""Fourth set of golden vectors (round 4): the REFERENCE's own look-ahead (abachurin/2048 @ /root/reference), run in the build
container with the chance nodes of the device's sampling spec — same import recipe as make_golden.py:

    python tests/golden/make_golden4.py

  lookahead_dev.npz    Game.look_forward (game_logic.py:214-243) and Game._find_best_move (:150-161) on 24 mid-game boards for
                       several (depth, width, since_empty), n = 3 and n = 5 dyadic tables; `random.sample` / `randrange(10)`
                       inside look_forward answer from 2048_amd/rng.py `lookahead_draws(board, k, salt)` — a function of the
                       node's board and a salt, which is what the device's level-by-level walk draws too.
  trial_lookahead.npz  QAgent.trial(depth=2, width=3, since_empty=9) (r_learning.py:348-406 -> Game.trial_run, game_logic.py:170-183)
                       of 4 games with the n = 4 dyadic table: game g draws its NEW TILES from lane LANE0 + g of the device RNG
                       spec and its chance nodes with salt = that lane's RNG state at the move.  Per game: score, odometer,
                       final row, starting position, moves, tiles.

Data only: inputs and the reference's outputs.
"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests.golden.make_golden import DrawShim, import_reference, save  # noqa: E402
from tests.golden.make_golden2 import dyadic_agent  # noqa: E402

rng_spec = importlib.import_module('2048_amd.rng')

LANE0, SEED, GAMES = 1 << 41, 4321, 4
TRIAL = dict(depth=2, width=3, since_empty=9)
CONFIGS = np.array([(1, 2, 16), (2, 3, 6), (3, 2, 8), (2, 4, 16), (3, 4, 6), (1, 16, 16)], np.int64)
SALTS = np.array([(0, 0), (0x0123456789ABCDEF, 0xFEDCBA9876543210)], np.uint64)


class TreeShim:
    """`random` for game_logic: inside look_forward, sample() and the randrange(10) calls behind it answer from
    rng.lookahead_draws of the board whose empty cells were just listed (Game.empty is wrapped to say which board that is);
    outside of it (Game.new_tile: randrange(10) then choice(empties)) the draws come from the current lane's stream."""

    def __init__(self, salt=(0, 0), seed=None, lane0=0):
        self.fixed_salt, self.seed, self.next_lane = salt, seed, lane0
        self.row, self.stack, self.lane = None, [], None

    def new_game(self):
        self.lane = DrawShim(self.seed, self.next_lane)
        self.next_lane += 1

    def note(self, row):
        self.row = np.array(row, np.uint8)

    def salt(self):
        return (self.lane.rng.s0, self.lane.rng.s1) if self.lane is not None else self.fixed_salt

    def sample(self, cells, k):
        draws = rng_spec.lookahead_draws(self.row.reshape(16).tolist(), k, self.salt())
        assert set((c >> 2, c & 3) for c, _ in draws) <= set((int(a), int(b)) for a, b in cells)
        if k:
            self.stack.append([t for _, t in draws])
        return [(c >> 2, c & 3) for c, _ in draws]

    def randrange(self, n):
        assert n == 10
        if self.stack:
            tile = self.stack[-1].pop(0)
            if not self.stack[-1]:
                self.stack.pop()
            return 0 if tile == 2 else 1
        return self.lane.randrange(n)

    def choice(self, seq):
        return self.lane.choice(seq)


def wrap_empty(gl, shim):
    real = gl.Game.empty

    def empty(row):
        shim.note(row)
        return real(row)
    gl.Game.empty = staticmethod(empty)
    return real


def lookahead_dev(gl, rl):
    g = np.load(os.path.join(HERE, 'moves.npz'))
    b = g['boards'][(g['game_over'] == 0) & (g['empty_count'] > 0) & (g['boards'].reshape(len(g['boards']), 16).max(axis=1) < 12)]
    boards = np.ascontiguousarray(b[::97][:24])
    out = {}
    for n, scale in ((3, 1.0), (5, 2.0 ** -6)):
        agent, _ = dyadic_agent(rl.QAgent, n, scale)
        values = np.zeros((len(SALTS), len(CONFIGS), len(boards)))
        best_dir = np.zeros((len(SALTS), len(CONFIGS), len(boards)), np.int8)
        for si, salt in enumerate(SALTS):
            shim = TreeShim(salt=(int(salt[0]), int(salt[1])))
            gl.random = shim
            real = wrap_empty(gl, shim)
            for ci, (depth, width, since_empty) in enumerate(CONFIGS):
                if n == 5 and depth == 3 and width == 4:
                    continue                                    # (minutes of Python per board; covered by n = 3)
                for bi, board in enumerate(boards):
                    game = gl.Game(row=board.astype(np.int32))
                    values[si, ci, bi] = game.look_forward(agent.evaluate, game.row, 0, depth=int(depth), width=int(width), since_empty=int(since_empty))
                    assert not shim.stack
                    best_dir[si, ci, bi] = game._find_best_move(agent.evaluate, int(depth), int(width), int(since_empty))[0]
                    assert not shim.stack
            gl.Game.empty = staticmethod(real)
        out[f'values_n{n}'], out[f'best_dir_n{n}'], out[f'scale_n{n}'] = values, best_dir, scale
        print(f'  n={n}: values in [{values.min():.3f}, {values.max():.3f}]')
    save('lookahead_dev.npz', boards=boards, configs=CONFIGS, salts=SALTS, **out)


def trial_lookahead(gl, rl):
    agent, _ = dyadic_agent(rl.QAgent, 4, 2.0 ** -6)
    shim = TreeShim(seed=SEED, lane0=LANE0)
    gl.random = shim
    real_empty = wrap_empty(gl, shim)
    real_init = gl.Game.__init__

    def init(self, score=0, row=None, file=None):
        if row is None:
            shim.new_game()
        real_init(self, score=score, row=row, file=file)
    gl.Game.__init__ = init
    lines = []
    rl.print = lambda *a, **k: lines.append(' '.join(str(x) for x in a))
    try:
        results = rl.QAgent.trial(estimator=agent.evaluate, num=GAMES, storage='local', console='local', **TRIAL)
    finally:
        gl.Game.__init__ = real_init
        gl.Game.empty = staticmethod(real_empty)
        del rl.print
    longest = max(len(g.moves) for g in results)
    moves = np.full((len(results), longest), -2, np.int8)
    tiles = np.zeros((len(results), longest, 2), np.uint8)
    for i, g in enumerate(results):
        assert len(g.moves) == len(g.tiles) == g.odometer
        moves[i, :len(g.moves)] = g.moves
        tiles[i, :len(g.tiles)] = [(t, p[0] * 4 + p[1]) for t, p in g.tiles]
    save('trial_lookahead.npz', n=4, seed=SEED, lane0=LANE0, scale=2.0 ** -6, depth=TRIAL['depth'], width=TRIAL['width'], since_empty=TRIAL['since_empty'],
         scores=np.array([g.score for g in results], np.int64), odometers=np.array([g.odometer for g in results], np.int64),
         rows=np.stack([g.row for g in results]).astype(np.uint8), starts=np.stack([g.starting_position for g in results]).astype(np.uint8),
         moves=moves, tiles=tiles)
    print(f'  trial(depth 2): scores {[g.score for g in results]}, moves {[g.odometer for g in results]}')


if __name__ == '__main__':
    gl, rl = import_reference()
    which = sys.argv[1:] or ['lookahead_dev', 'trial_lookahead']
    for name in which:
        print(name)
        globals()[name](gl, rl)
