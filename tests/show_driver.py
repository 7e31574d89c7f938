"""BASELINE config 1 — "show.py still drives it" — as something a test can run where the reference is absent.

tests/golden/make_show_transcript.py runs the REFERENCE's own show.py (options 1, 2, 3; headless, scripted input, tests/pygame_stub.py
as `pygame`) on top of this repository's `game2048` package and stores what it drew as tests/golden/show_transcript.npz.
This module holds the two things that script and the tests share:

  * `prepare_storage`  the local store show.py's menu reads (an agent in the two-object 's3' form, one saved game);
  * `drive`            the same sequence of SURFACE calls show.py makes for each option, restated on the bare surface
                       (show.py:104-149 `Show.replay` / `Show.watch`, :152-168 `input_name`, :199-216 the menu) and producing
                       the same frame records — so that `-m gpu` can put the HIP backend through the call sequence and
                       compare with what the reference's script drew on the CPU backend, without the reference on the box.

Frame record: [score, moves, move or -1, over, 16 face values row-major] (pygame_stub.Recorder.frames).
"""
import importlib
import random

import numpy as np

from tests.golden import formulas

AGENT_NAME, AGENT_N, SCALE = 'show_agent', 4, 2.0 ** -6
SEEDS = {1: 101, 2: 202, 3: 303}                    # `random.seed` before each option (Show.__init__ and watch draw tiles from it)


def set_storage(path):
    """Point the local store (2048_amd/start.py: load_s3 / save_s3 / list_names_s3) at `path`."""
    importlib.import_module('2048_amd.start').STORAGE = str(path)


def prepare_storage(rl, path):
    """An n = 4 agent with the dyadic test table, saved the way QAgent.save_agent does for storage='s3' (a/<name>.pkl +
    weights/<name>.pkl, r_learning.py:166-176), and the best of four of its games under g/ (what option 1 lists)."""
    set_storage(path)
    agent = rl.QAgent(name=AGENT_NAME, storage='s3', console='local', n=AGENT_N, with_weights=False)
    sizes = formulas.feature_sizes(AGENT_N)
    flat = formulas.weights(AGENT_N, scale=SCALE).astype(np.float32)
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]])
    agent.weights = [flat[o:o + s] for o, s in zip(offs, sizes)]
    agent.save_agent()
    agent.trial_seed = (77, 1 << 42)
    quiet(lambda: rl.QAgent.trial(estimator=agent.evaluate, num=4, storage='s3', console='local', game_file='g/' + agent.game_file))
    return sorted(rl.list_names_s3())


def quiet(fn):
    import builtins
    real = builtins.print
    builtins.print = lambda *a, **k: None
    try:
        return fn()
    finally:
        builtins.print = real


def face(row):
    return [(1 << int(v)) if v else 0 for v in np.asarray(row).reshape(16)]


def replay_frames(rl, shown):
    """What Show.replay paints (show.py:104-118): the recorded game re-run move by move from its starting position."""
    board = rl.Game(row=shown.starting_position)
    frames = []
    for i in range(shown.odometer):
        move = shown.moves[i]
        tile, cell = shown.tiles[i]
        frames.append([board.score, board.odometer, move, 0] + face(board.row))
        board.make_move(move)
        board.row[cell] = tile
    frames.append([board.score, board.odometer, -1, 1] + face(board.row))
    return frames


def watch_frames(rl, estimator):
    """What Show.watch paints (show.py:128-148): Game.generate_run yields (game, move) before each move; then "Over!"."""
    game = rl.Game()
    frames = []
    for state, move in game.generate_run(estimator=estimator, depth=0, width=1, since_empty=6):
        frames.append([state.score, state.odometer, move, 0] + face(state.row))
    frames.append([game.score, game.odometer, -1, 1] + face(game.row))
    return frames


def pick(rl, what):
    """input_name (show.py:152-168): the entries of the store whose key starts with the kind's letter."""
    items = {i: v for i, v in enumerate(rl.list_names_s3()) if v[:2] == f'{what[0]}/'}
    idx = sorted(items)[0]
    return idx, (rl.load_s3(items[idx]) if what == 'game' else rl.QAgent.load_agent(items[idx]))


def drive(rl, option):
    """The menu's branch `option` (show.py:199-216) on the bare surface; returns (frames, scores of the trial or [])."""
    random.seed(SEEDS[option])
    rl.Game()                                       # Show.__init__ starts a game of its own (show.py:35): two tiles' worth of draws
    if option == 1:
        _, game = pick(rl, 'game')
        return replay_frames(rl, game), []
    _, agent = pick(rl, 'agent')
    est = agent.evaluate
    if option == 2:
        results = quiet(lambda: rl.QAgent.trial(estimator=est, num=100, console='local'))
        return replay_frames(rl, results[0]), [g.score for g in results]
    return watch_frames(rl, est), []
