#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE itself (abachurin/2048 @ /root/reference).

Run in the build container only (the reference never travels to the GPU box):

    python tests/golden/make_golden.py

Outputs `tests/golden/*.npz` — data only (inputs and the reference's outputs).  The reference is
imported with an empty stub for the absent `boto3` module and S3_URL=none, so that
game2048/start.py:35-51 takes its "unknown environment" branch and touches no storage.
Spawn draws are injected by replacing the module-global `random` that game_logic.py resolves
(start.py:6) with a shim that follows the device RNG spec (2048_amd/rng.py).
"""
import importlib
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
rng_spec = importlib.import_module('2048_amd.rng')
from tests.golden import formulas  # noqa: E402


def import_reference():
    os.environ['S3_URL'] = 'none'
    sys.modules.setdefault('boto3', types.ModuleType('boto3'))
    sys.path.insert(0, '/root/reference')
    import game2048.game_logic as gl
    import game2048.r_learning as rl
    return gl, rl


class DrawShim:
    """Stands in for the `random` module inside game_logic: one xoroshiro draw per new tile."""

    def __init__(self, seed, lane=0):
        self.rng = rng_spec.LaneRng(seed, lane)
        self.pending = None
        self.log = []

    def randrange(self, n):
        assert n == 10
        self.pending = self.rng.next()
        return ((self.pending >> 32) * 10) >> 32

    def choice(self, seq):
        r10, k = rng_spec.spawn_draw(self.pending, len(seq))
        self.log.append((r10, k, len(seq)))
        return seq[k]


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f'{name}: {os.path.getsize(path) / 1024:.1f} KiB')


def main():
    gl, rl = import_reference()
    Game, QAgent = gl.Game, rl.QAgent

    # ---- 1. move table (game_logic.py:18-39)
    out = np.zeros((65536, 4), np.uint8)
    score = np.zeros(65536, np.uint32)
    changed = np.zeros(65536, np.uint8)
    for key in range(65536):
        line = ((key >> 12) & 15, (key >> 8) & 15, (key >> 4) & 15, key & 15)
        o, s, c = Game.table[line]
        out[key], score[key], changed[key] = o, s, c
    save('move_table.npz', out=out, score=score, changed=changed)

    # ---- 2/3. boards: pre_move x4, game_over, empty_count, adjacent_pair_count
    import random as pyrandom
    gl.random = pyrandom.Random(7)                    # seeded stand-in for the module-global `random`
    played = []
    for _ in range(24):                               # boards from actual random play of the reference Game
        game = Game()
        while not game.game_over(game.row):
            if game.make_move(gl.random.randrange(4)):
                game.new_tile()
            if game.odometer % 3 == 0:
                played.append(game.row.copy())
        played.append(game.row.copy())
    boards = formulas.fixture_boards(played)
    N = len(boards)
    g = Game(row=np.zeros((4, 4), np.int32))
    after = np.zeros((N, 4, 4, 4), np.uint8)
    reward = np.zeros((N, 4), np.int32)
    moved = np.zeros((N, 4), np.uint8)
    over = np.zeros(N, np.uint8)
    n_empty = np.zeros(N, np.uint8)
    n_pairs = np.zeros(N, np.uint8)
    for i, b in enumerate(boards):
        row = b.astype(np.int32)
        for d in range(4):
            nr, ns, ch = g.pre_move(row, 0, d)
            after[i, d], reward[i, d], moved[i, d] = nr, ns, ch
        over[i] = g.game_over(row)
        n_empty[i] = Game.empty_count(row)
        n_pairs[i] = Game.adjacent_pair_count(row)
    save('moves.npz', boards=boards, after=after, reward=reward, changed=moved, game_over=over,
         empty_count=n_empty, adjacent_pair_count=n_pairs)

    # ---- 4. spawn with injected draws (game_logic.py:112-121)
    r = np.random.RandomState(2048)
    sp_idx = np.nonzero(n_empty > 0)[0]
    sp_boards = boards[sp_idx]
    sp_r10 = r.randint(0, 10, len(sp_idx)).astype(np.uint8)
    sp_k = (r.randint(0, 1 << 30, len(sp_idx)) % n_empty[sp_idx]).astype(np.uint8)
    sp_after = np.zeros_like(sp_boards)

    class Inject:
        def __init__(self):
            self.r10 = self.k = None

        def randrange(self, n):
            return int(self.r10)

        def choice(self, seq):
            return seq[int(self.k)]
    inj = Inject()
    gl.random = inj
    for i in range(len(sp_idx)):
        game = Game(row=sp_boards[i].astype(np.int32))
        inj.r10, inj.k = sp_r10[i], sp_k[i]
        game.new_tile()
        sp_after[i] = game.row
    save('spawn.npz', boards=sp_boards, r10=sp_r10, k=sp_k, after=sp_after)

    # ---- 5. feature indices f_2..f_6 (r_learning.py:17-69)
    feats = {}
    for n in range(2, 7):
        fn = QAgent.feature_functions[n]
        feats[f'f{n}'] = np.stack([fn(b.astype(np.int32)) for b in boards]).astype(np.int32)
    save('features.npz', boards=boards, **feats)

    # ---- 6. evaluate / greedy select / update with exactly representable weights
    ev = {}
    # afterstates must stay inside the reference's tile domain 0..15 (a 15+15 merge makes a 16, which
    # QAgent.evaluate cannot index): keep boards whose tiles are <= 14 for the greedy-choice vectors
    ok = (over == 0) & moved.any(axis=1) & (boards.reshape(N, 16).max(axis=1) <= 14)
    sel_boards = boards[ok][::2][:1024]
    up_states = boards[::37][:96]
    up_dw = formulas.update_dws(len(up_states))
    for n in range(2, 7):
        agent = QAgent(name='golden', storage='local', console='local', n=n, with_weights=False)
        sizes = formulas.feature_sizes(n)
        flat = formulas.weights(n).astype(np.float64)
        offs = np.concatenate([[0], np.cumsum(sizes)[:-1]])
        agent.weights = [flat[o:o + s] for o, s in zip(offs, sizes)]       # rows are views of `flat`
        ev[f'value{n}'] = np.array([agent.evaluate(b.astype(np.int32)) for b in boards], np.float64)
        # greedy choice exactly as Game._find_best_move with depth 0 (game_logic.py:150-161)
        act = np.zeros(len(sel_boards), np.uint8)
        for i, b in enumerate(sel_boards):
            game = Game(row=b.astype(np.int32))
            best_dir, best_row, best_score = game._find_best_move(agent.evaluate, 0, 1, 0)
            act[i] = best_dir
        ev[f'action{n}'] = act
        # update (r_learning.py:207-214): sparse difference of the table
        before = flat.copy()
        for s, dw in zip(up_states, up_dw):
            agent.update(s.astype(np.int32), float(dw))
        diff = flat - before
        nz = np.nonzero(diff)[0]
        ev[f'upd_slot{n}'] = nz.astype(np.int64)
        ev[f'upd_delta{n}'] = diff[nz]
        del agent, flat, before, diff
    save('learner.npz', boards=boards, sel_boards=sel_boards, up_states=up_states, up_dw=up_dw, **ev)

    # ---- 7. full episode() traces with injected draws (r_learning.py:224-252)
    for n, seed in ((2, 11), (3, 12), (4, 13)):
        F = QAgent.parameter_shape[n][0]
        alpha = formulas.exact_alpha(n)
        agent = QAgent(name='golden', storage='local', console='local', n=n, alpha=alpha, with_weights=False)
        sizes = formulas.feature_sizes(n)
        flat = formulas.weights(n, scale=2.0 ** -6).astype(np.float64)
        offs = np.concatenate([[0], np.cumsum(sizes)[:-1]])
        agent.weights = [flat[o:o + s] for o, s in zip(offs, sizes)]
        before = flat.copy()
        shim = DrawShim(seed)
        gl.random = shim
        # instrument update() to record (state, dw) without changing what it does
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
        tr_boards = np.stack([replay[i][0] for i in range(steps + 1)]).astype(np.uint8)
        tr_scores = np.array([replay[i][1] for i in range(steps + 1)], np.int64)
        tr_moves = np.array(game.moves, np.int8)
        tiles = np.array([(t, p[0] * 4 + p[1]) for t, p in game.tiles], np.uint8)
        save(f'episode_n{n}.npz', n=n, seed=seed, alpha=alpha, start=np.array(game.starting_position, np.uint8),
             boards=tr_boards, scores=tr_scores, moves=tr_moves, tiles=tiles,
             draws=np.array(shim.log, np.int64), rec_states=np.stack(rec_states), rec_dw=np.array(rec_dw),
             w_slot=nz.astype(np.int64), w_delta=diff[nz], final_score=game.score, final_board=game.row.astype(np.uint8))
        print(f'  episode n={n}: {steps} moves, score {game.score}')


if __name__ == '__main__':
    main()
