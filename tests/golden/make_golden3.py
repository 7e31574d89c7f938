#!/usr/bin/env python3
"""Third set of golden vectors (round 3), again produced by running the REFERENCE itself (abachurin/2048 @
/root/reference) in the build container — same import recipe as make_golden.py (stub for the absent boto3, S3_URL=none):

    python tests/golden/make_golden3.py            # after  python tests/golden/make_built_pickles.py

  trial.npz           QAgent.trial (r_learning.py:348-406) -> Game.trial_run (game_logic.py:170-183), depth 0, 8 games with
                      dyadic n=4 weights; game g draws its tiles from lane LANE0 + g of the device RNG spec.  Per game (in
                      the order trial returns them: best first): score, odometer, final row, starting position, moves, tiles.
  trial_init.npz      the same with game_init = a mid-game position (r_learning.py:363: game_init.copy()), 6 games.
  built_pickles.npz   what the REFERENCE makes of pickles written by THIS build (tests/golden/built_*.pkl, produced by
                      make_built_pickles.py with 2048_amd's own classes): the agent's evaluate() on golden boards after the
                      reference's own load path (pickle.load + np_to_list, r_learning.py:189-200), attribute values, and the
                      chain Game.replay() (game_logic.py:246-269) rebuilds from the build's Game record.

Data only: inputs and the reference's outputs.
"""
import os
import pickle
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests.golden import formulas  # noqa: E402
from tests.golden.make_golden import DrawShim, import_reference, save  # noqa: E402
from tests.golden.make_golden2 import dyadic_agent  # noqa: E402

TRIAL_SEED, TRIAL_LANE0, TRIAL_GAMES, TRIAL_N = 2125, 1 << 41, 8, 4


class LaneShim:
    """`random` for game_logic during QAgent.trial: every new game (Game() without a row) moves on to the next lane's
    xoroshiro stream, as the device's lanes do."""

    def __init__(self, seed, lane0):
        self.seed, self.next_lane, self.cur = seed, lane0, None

    def new_game(self):
        self.cur = DrawShim(self.seed, self.next_lane)
        self.next_lane += 1

    def randrange(self, n):
        return self.cur.randrange(n)

    def choice(self, seq):
        return self.cur.choice(seq)


def trial(gl, rl):
    agent, _ = dyadic_agent(rl.QAgent, TRIAL_N, 2.0 ** -6)
    shim = LaneShim(TRIAL_SEED, TRIAL_LANE0)
    gl.random = shim
    real_init = gl.Game.__init__

    def init(self, score=0, row=None, file=None):
        if row is None:
            shim.new_game()
        real_init(self, score=score, row=row, file=file)
    gl.Game.__init__ = init
    lines = []
    import builtins
    real_print = builtins.print
    rl.print = lambda *a, **k: lines.append(' '.join(str(x) for x in a))       # (trial's display IS print for console='local')
    try:
        results = rl.QAgent.trial(estimator=agent.evaluate, num=TRIAL_GAMES, storage='local', console='local')
    finally:
        gl.Game.__init__ = real_init
        del rl.print
    longest = max(len(g.moves) for g in results)
    moves = np.full((len(results), longest), -2, np.int8)
    tiles = np.zeros((len(results), longest, 2), np.uint8)
    for i, g in enumerate(results):
        assert len(g.moves) == len(g.tiles) == g.odometer
        moves[i, :len(g.moves)] = g.moves
        tiles[i, :len(g.tiles)] = [(t, p[0] * 4 + p[1]) for t, p in g.tiles]
    save('trial.npz', n=TRIAL_N, seed=TRIAL_SEED, lane0=TRIAL_LANE0, scale=2.0 ** -6,
         scores=np.array([g.score for g in results], np.int64), odometers=np.array([g.odometer for g in results], np.int64),
         rows=np.stack([g.row for g in results]).astype(np.uint8),
         starts=np.stack([g.starting_position for g in results]).astype(np.uint8), moves=moves, tiles=tiles,
         summary=np.array('\n'.join(lines)))
    real_print(f'  trial: scores {[g.score for g in results]}, moves {[g.odometer for g in results]}')


def trial_with_game_init(gl, rl):
    """QAgent.trial(game_init=...) (r_learning.py:362-365): every game is `game_init.copy()` = Game(score, row) — a FRESH record
    (odometer 0, no moves, no tiles, starting_position = game_init.row) that continues from a mid-game position.  Game g draws
    its tiles from lane LANE0 + g of the RNG spec, behind the two draws a fresh game of that lane would have used for its
    first tiles (the device seeds a lane by starting a game in it; the position is then set from game_init)."""
    agent, _ = dyadic_agent(rl.QAgent, TRIAL_N, 2.0 ** -6)
    boards = np.load(os.path.join(HERE, 'features.npz'))['boards']
    pick = next(b for b in boards if b.max() == 7 and (b == 0).sum() >= 4)        # a mid-game position, some room left
    game_init = gl.Game(score=1500, row=pick.astype(np.int32))
    game_init.odometer, game_init.moves, game_init.tiles = 3, [0, 1, 2], [(1, (0, 0))] * 3          # a prefix the copies must NOT inherit
    shim = LaneShim(TRIAL_SEED + 1, TRIAL_LANE0)
    gl.random = shim
    real_init = gl.Game.__init__

    def init(self, score=0, row=None, file=None):
        if row is not None:                                 # game_init.copy()
            shim.new_game()
            for cells in (16, 15):                          # the lane's own first two tiles, not used
                shim.randrange(10)
                shim.choice(list(range(cells)))
        real_init(self, score=score, row=row, file=file)
    gl.Game.__init__ = init
    lines = []
    rl.print = lambda *a, **k: lines.append(' '.join(str(x) for x in a))
    try:
        results = rl.QAgent.trial(estimator=agent.evaluate, num=6, game_init=game_init, storage='local', console='local')
    finally:
        gl.Game.__init__ = real_init
        del rl.print
    longest = max(len(g.moves) for g in results)
    moves = np.full((len(results), longest), -2, np.int8)
    tiles = np.zeros((len(results), longest, 2), np.uint8)
    for i, g in enumerate(results):
        assert len(g.moves) == len(g.tiles) == g.odometer
        moves[i, :len(g.moves)] = g.moves
        tiles[i, :len(g.tiles)] = [(t, p[0] * 4 + p[1]) for t, p in g.tiles]
    save('trial_init.npz', n=TRIAL_N, seed=TRIAL_SEED + 1, lane0=TRIAL_LANE0, scale=2.0 ** -6, init_row=pick.astype(np.uint8), init_score=1500,
         scores=np.array([g.score for g in results], np.int64), odometers=np.array([g.odometer for g in results], np.int64),
         rows=np.stack([g.row for g in results]).astype(np.uint8),
         starts=np.stack([np.asarray(g.starting_position) for g in results]).astype(np.uint8), moves=moves, tiles=tiles,
         summary=np.array('\n'.join(lines)))
    print(f'  trial(game_init): scores {[g.score for g in results]}, moves {[g.odometer for g in results]}')


def built_pickles(gl, rl):
    """The reverse direction of f-2: objects pickled by the build, read by the reference."""
    g = np.load(os.path.join(HERE, 'features.npz'))
    boards = g['boards'][:64].astype(np.int32)
    out = {}
    # local form: the whole agent in one pickle; the reference's load path is pickle.load + np_to_list (r_learning.py:189-193;
    # its open(file, 'r') is a bug of its own: binary mode here)
    with open(os.path.join(HERE, 'built_agent_local.pkl'), 'rb') as f:
        agent = pickle.load(f)
    assert type(agent) is rl.QAgent, type(agent)
    agent.np_to_list()
    out['local_values'] = np.array([agent.evaluate(b) for b in boards])
    out['local_attrs'] = np.array([agent.n, agent.num_feat, agent.step, agent.top_score, agent.top_tile, agent.decay_step], np.int64)
    out['local_alpha'] = np.array([agent.alpha, agent.decay, agent.low_alpha_limit])
    out['local_signature'] = np.array(agent.weight_signature, np.int64)
    out['local_history'] = np.array(agent.train_history, np.int64)
    before = agent.evaluate(boards[5])
    agent.update(boards[5], 0.125)                              # and the reference can keep training it
    out['local_update_gain'] = np.array(agent.evaluate(boards[5]) - before)
    # s3 form: parameters and weights apart (r_learning.py:196-200)
    with open(os.path.join(HERE, 'built_agent_params.pkl'), 'rb') as f:
        params = pickle.load(f)
    with open(os.path.join(HERE, 'built_agent_weights.pkl'), 'rb') as f:
        params.weights = pickle.load(f)
    params.np_to_list()
    out['s3_values'] = np.array([params.evaluate(b) for b in boards])
    out['s3_name'] = np.array(params.name)
    # a Game record
    game = gl.Game.load_game(os.path.join(HERE, 'built_game.pkl'))
    assert type(game) is gl.Game
    chain = game.replay(verbose=False)
    out['game_chain_rows'] = np.stack([chain[i][0] for i in range(game.odometer + 1)]).astype(np.uint8)
    out['game_chain_scores'] = np.array([chain[i][1] for i in range(game.odometer + 1)], np.int64)
    out['game_chain_moves'] = np.array([chain[i][2] for i in range(game.odometer + 1)], np.int64)
    out['game_str'] = np.array(str(game))
    save('built_pickles.npz', boards=boards.astype(np.uint8), **out)


if __name__ == '__main__':
    gl, rl = import_reference()
    trial(gl, rl)
    trial_with_game_init(gl, rl)
    if os.path.exists(os.path.join(HERE, 'built_agent_local.pkl')):
        built_pickles(gl, rl)
    else:
        print('built_*.pkl missing: run tests/golden/make_built_pickles.py first')
