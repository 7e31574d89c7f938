#!/usr/bin/env python3
"""Second set of golden vectors, again produced by running the REFERENCE itself (abachurin/2048 @ /root/reference) in
the build container — same import recipe as make_golden.py (stub for the absent boto3, S3_URL=none):

    python tests/golden/make_golden2.py

  episode_n5.npz, episode_n6.npz   whole QAgent.episode() traces (r_learning.py:224-252), injected draws
  lookahead.npz                    Game.look_forward / _find_best_move values (game_logic.py:150-161,214-243)
  train_schedule.npz               QAgent.train_run (r_learning.py:254-346): alpha / decay schedule, ma100 history, log
  ref_agent_local.pkl              a reference agent saved in local mode (r_learning.py:176-180)
  ref_agent_params.pkl, ref_agent_weights.pkl   the two objects save_agent hands to save_s3 in s3 mode (:166-175)
  ref_game.pkl                     a reference Game saved with Game.save_game (game_logic.py:77-80)
  ref_pickles.npz                  what the loaded objects must show (values, attributes)

Data only: inputs and the reference's outputs.  Pickles written by the reference carry class PATHS
(game2048.r_learning.QAgent, game2048.game_logic.Game), no code.
"""
import io
import os
import pickle
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests.golden import formulas  # noqa: E402
from tests.golden.make_golden import DrawShim, import_reference, save  # noqa: E402


def dyadic_agent(QAgent, n, scale, **kw):
    agent = QAgent(name='golden', storage='local', console='local', n=n, with_weights=False, **kw)
    sizes = formulas.feature_sizes(n)
    flat = formulas.weights(n, scale=scale).astype(np.float64)
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]])
    agent.weights = [flat[o:o + s] for o, s in zip(offs, sizes)]       # rows are views of `flat`
    return agent, flat


def episodes(gl, rl):
    for n, seed in ((5, 14), (6, 15)):
        alpha = formulas.exact_alpha(n)
        agent, flat = dyadic_agent(rl.QAgent, n, 2.0 ** -6, alpha=alpha)
        before = flat.copy()
        shim = DrawShim(seed)
        gl.random = shim
        rec_states, rec_dw = [], []
        real_update = agent.update

        def spy(row, dw, _u=real_update):
            rec_states.append(np.array(row, np.uint8))
            rec_dw.append(dw)
            _u(row, dw)
        agent.update = spy
        game = agent.episode()
        diff = flat - before
        nz = np.nonzero(diff)[0]
        replay = game.replay(verbose=False)
        steps = game.odometer
        save(f'episode_n{n}.npz', n=n, seed=seed, alpha=alpha, start=np.array(game.starting_position, np.uint8),
             boards=np.stack([replay[i][0] for i in range(steps + 1)]).astype(np.uint8),
             scores=np.array([replay[i][1] for i in range(steps + 1)], np.int64), moves=np.array(game.moves, np.int8),
             tiles=np.array([(t, p[0] * 4 + p[1]) for t, p in game.tiles], np.uint8),
             draws=np.array(shim.log, np.int64), rec_states=np.stack(rec_states), rec_dw=np.array(rec_dw),
             w_slot=nz.astype(np.int64), w_delta=diff[nz], final_score=game.score, final_board=game.row.astype(np.uint8))
        print(f'  episode n={n}: {steps} moves, score {game.score}')
        del agent, flat, before, diff


class BoardKeyedShim:
    """`random` for look_forward: sample() and the randrange() calls that follow it answer from formulas.lookahead_draws
    of the board whose empty cells were just listed (Game.empty is wrapped to tell us which board that is)."""

    def __init__(self):
        self.row = None
        self.stack = []

    def note(self, row):
        self.row = np.array(row, np.uint8)

    def sample(self, cells, k):
        flat, tiles = formulas.lookahead_draws(self.row, k)
        assert set((int(c) >> 2, int(c) & 3) for c in flat) <= set((int(a), int(b)) for a, b in cells)
        self.stack.append(list(tiles))
        return [(int(c) >> 2, int(c) & 3) for c in flat]

    def randrange(self, n):
        assert n == 10
        tile = self.stack[-1].pop(0)
        if not self.stack[-1]:
            self.stack.pop()
        return 0 if tile == 2 else 1


def lookahead(gl, rl):
    n = 3
    agent, _ = dyadic_agent(rl.QAgent, n, 1.0)
    g = np.load(os.path.join(HERE, 'moves.npz'))
    b = g['boards'][(g['game_over'] == 0) & (g['empty_count'] > 0) & (g['boards'].reshape(len(g['boards']), 16).max(axis=1) < 12)]
    boards = np.ascontiguousarray(b[::97][:24])
    shim = BoardKeyedShim()
    gl.random = shim
    real_empty = gl.Game.empty

    def empty(row):
        shim.note(row)
        return real_empty(row)
    gl.Game.empty = staticmethod(empty)
    configs = np.array([(1, 2, 16), (2, 3, 6), (3, 2, 8), (2, 4, 16)], np.int64)
    values = np.zeros((len(configs), len(boards)))
    best_dir = np.zeros((len(configs), len(boards)), np.int8)
    for ci, (depth, width, since_empty) in enumerate(configs):
        for bi, board in enumerate(boards):
            game = gl.Game(row=board.astype(np.int32))
            values[ci, bi] = game.look_forward(agent.evaluate, game.row, 0, depth=int(depth), width=int(width), since_empty=int(since_empty))
            assert not shim.stack
            best_dir[ci, bi] = game._find_best_move(agent.evaluate, int(depth), int(width), int(since_empty))[0]
            assert not shim.stack
    gl.Game.empty = staticmethod(real_empty)
    save('lookahead.npz', n=n, boards=boards, configs=configs, values=values, best_dir=best_dir)


def train_schedule(gl, rl):
    """1 010 episodes of the reference's train_run with n = 2 (r_learning.py:269-346) under a short decay_step, the
    spawn draws from the device RNG spec.  What is recorded is what the schedule sees and does: every episode's outcome
    (final row, score) and the learning rate in force when it started; the ma100 history; the log text."""
    params = dict(n=2, alpha=0.2, decay=0.75, decay_step=150, low_alpha_limit=0.02)
    np.random.seed(5)
    agent = rl.QAgent(name='sched', storage='local', console='local', **params)
    agent.top_tile = 6                                        # so that "new maximum tile" decays (:311-313) happen for a novice
    gl.random = DrawShim(77)
    lines = []
    agent.print = lambda text='': lines.append(str(text))
    rows, scores, odometers, alpha_at_start, step_at_start, next_decay_at_start = [], [], [], [], [], []
    real_episode = agent.episode

    def episode():
        alpha_at_start.append(agent.alpha)
        step_at_start.append(agent.step)
        next_decay_at_start.append(agent.next_decay)
        game = real_episode()
        rows.append(game.row.astype(np.uint8))
        scores.append(game.score)
        odometers.append(game.odometer)
        return game
    agent.episode = episode
    agent.train_run(num_eps=1009, saving=False)
    keep = [ln for ln in lines if not ln.rstrip().endswith(' min') and not ln.startswith('Total time')]
    save('train_schedule.npz', rows=np.stack(rows), scores=np.array(scores, np.int64), odometers=np.array(odometers, np.int64), alpha_at_start=np.array(alpha_at_start),
         step_at_start=np.array(step_at_start, np.int64), next_decay_at_start=np.array(next_decay_at_start, np.int64),
         train_history=np.array(agent.train_history, np.int64), final_alpha=agent.alpha, final_step=agent.step,
         final_top_tile=agent.top_tile, final_top_score=agent.top_score, final_next_decay=agent.next_decay,
         top_game_score=agent.top_game.score, params=np.array([params[k] for k in ('n', 'alpha', 'decay', 'decay_step', 'low_alpha_limit')]),
         log=np.array('\n'.join(keep)))
    print(f'  train_run: {len(scores)} episodes, alpha {params["alpha"]} -> {agent.alpha}, top tile {agent.top_tile}, {len(keep)} log lines')


def pickles(gl, rl):
    np.random.seed(11)
    gl.random = DrawShim(5)
    agent = rl.QAgent(name='ref_agent', storage='local', console='local', n=2, alpha=0.1)
    for _ in range(3):
        game = agent.episode()
    agent.top_game, agent.top_score = game, game.score
    agent.train_history = [100, 200]
    boards = np.load(os.path.join(HERE, 'moves.npz'))['boards'][::211][:16]
    values = np.array([agent.evaluate(b.astype(np.int32)) for b in boards])
    groups = agent.list_to_np()
    # local mode, r_learning.py:176-180: weights as float32 groups inside the pickled agent
    cwd = os.getcwd()
    os.chdir(HERE)
    try:
        agent.file = 'ref_agent_local.pkl'
        agent.save_agent()                                    # (s3 is False: pickle.dump(self, f, -1), then np_to_list)
        # s3 mode, r_learning.py:166-175: the two objects that go to save_s3 (pickle.dump(data, f, -1), start.py:112-114)
        params = rl.QAgent(name=agent.name, with_weights=False)
        for key in agent.__dict__:
            if key != 'weights':
                setattr(params, key, getattr(agent, key))
        with open('ref_agent_params.pkl', 'wb') as f:
            pickle.dump(params, f, -1)
        with open('ref_agent_weights.pkl', 'wb') as f:
            pickle.dump(agent.list_to_np(), f, -1)
        game.save_game('ref_game.pkl')                        # game_logic.py:77-80
    finally:
        os.chdir(cwd)
    save('ref_pickles.npz', boards=boards, values=values, weights=np.concatenate([g.reshape(-1) for g in groups]),
         step=agent.step, alpha=agent.alpha, top_score=agent.top_score, train_history=np.array(agent.train_history),
         game_row=game.row.astype(np.uint8), game_score=game.score, game_moves=np.array(game.moves, np.int8),
         game_tiles=np.array([(t, p[0] * 4 + p[1]) for t, p in game.tiles], np.uint8),
         game_start=np.array(game.starting_position, np.uint8))


def main():
    gl, rl = import_reference()
    which = sys.argv[1:] or ['episodes', 'lookahead', 'train_schedule', 'pickles']
    for name in which:
        print(name)
        globals()[name](gl, rl)


if __name__ == '__main__':
    main()
