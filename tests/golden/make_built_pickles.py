#!/usr/bin/env python3
"""Pickles written by THIS build (2048_amd's own QAgent / Game classes, no GPU needed: the weights are handed to the agent
before any device use), for the reverse direction of SURVEY.md 8(f-2): tests/golden/make_golden3.py then loads them with
the reference and records what it sees.

    python tests/golden/make_built_pickles.py

  built_agent_local.pkl                           QAgent.save_agent in local mode (r_learning.py:176-180)
  built_agent_params.pkl, built_agent_weights.pkl the two objects save_agent hands to save_s3 in s3 mode (:166-175)
  built_game.pkl                                  Game.save_game (game_logic.py:77-80) of the n=2 golden episode's record
"""
import os
import pickle
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
os.environ.setdefault('S3_URL', 'none')
from tests.golden import formulas  # noqa: E402
import game2048.r_learning as rl  # noqa: E402  (the build's alias package: same class paths as the reference's)
import game2048.game_logic as gl  # noqa: E402

N = 3


def agent_with(name, storage):
    agent = rl.QAgent(name=name, storage=storage, console='local', n=N, alpha=0.125, decay=0.5, decay_step=777, low_alpha_limit=0.03,
                      with_weights=False)
    sizes = formulas.feature_sizes(N)
    flat = formulas.weights(N, scale=2.0 ** -4).astype(np.float32)
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]])
    agent.weights = [flat[o:o + s] for o, s in zip(offs, sizes)]
    agent.step, agent.top_score, agent.top_tile, agent.train_history = 4321, 98765, 12, [100, 250, 400]
    return agent


def main():
    cwd = os.getcwd()
    os.chdir(HERE)
    try:
        local = agent_with('built_agent_local', 'local')
        local.save_agent()                                     # -> built_agent_local.pkl
        s3 = agent_with('built_s3', 's3')
        captured = {}
        import importlib
        agent_mod = importlib.import_module('2048_amd.agent')
        real = agent_mod.save_s3
        agent_mod.save_s3 = lambda obj, name: captured.__setitem__(name, pickle.dumps(obj, -1))
        try:
            s3.save_agent()
        finally:
            agent_mod.save_s3 = real
        with open('built_agent_params.pkl', 'wb') as f:
            f.write(captured['a/built_s3.pkl'])
        with open('built_agent_weights.pkl', 'wb') as f:
            f.write(captured['weights/built_s3.pkl'])
        ep = np.load('episode_n2.npz')
        game = gl.Game(score=int(ep['final_score']), row=ep['final_board'].astype(np.int32))
        game.starting_position = ep['start'].astype(np.int32)
        game.moves = [int(m) for m in ep['moves']]
        game.tiles = [(int(t), (int(c) >> 2, int(c) & 3)) for t, c in ep['tiles']]
        game.odometer = len(game.tiles)
        game.save_game('built_game.pkl')
    finally:
        os.chdir(cwd)
    for f in ('built_agent_local.pkl', 'built_agent_params.pkl', 'built_agent_weights.pkl', 'built_game.pkl'):
        print(f, os.path.getsize(os.path.join(HERE, f)), 'bytes')


if __name__ == '__main__':
    main()
