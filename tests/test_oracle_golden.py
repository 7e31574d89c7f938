"""Pin the oracle (oracle/ref_scalar.py, oracle/ref_batch.py) to vectors produced by the reference
itself (tests/golden/make_golden.py).  CPU only."""
import hashlib

import numpy as np
import pytest

from oracle import ref_batch as rb
from oracle import ref_scalar as rs
from tests.golden import formulas


def test_move_table_matches_reference(golden):
    g = golden('move_table.npz')
    out, score, changed = rb.row_lut()
    assert np.array_equal(out, g['out'])
    assert np.array_equal(score, g['score'])
    assert np.array_equal(changed, g['changed'].astype(bool))
    # SURVEY.md §8c known answer
    h = hashlib.sha256()
    for k in range(65536):
        h.update(bytes(out[k].tolist()) + int(score[k]).to_bytes(4, 'little') + bytes([int(changed[k])]))
    assert h.hexdigest() == '335fb5cd3e92a895769af3f58d6be20cb3d05d9bf83f8434758adb809db2803e'
    assert changed.sum() == 21210 and score.sum() == 100660224


def test_moves_and_terminal(golden):
    g = golden('moves.npz')
    boards = g['boards']
    after, reward, changed = rb.move_all(boards)
    assert np.array_equal(after, g['after'])
    assert np.array_equal(reward, g['reward'])
    assert np.array_equal(changed, g['changed'].astype(bool))
    assert np.array_equal(rb.game_over(boards), g['game_over'].astype(bool))
    assert np.array_equal(rb.empty_count(boards), g['empty_count'])
    assert np.array_equal(rb.adjacent_pair_count(boards), g['adjacent_pair_count'])
    # scalar restatement on a subset
    for i in range(0, len(boards), 17):
        b = boards[i].astype(np.int32)
        for d in range(4):
            nb, s, c = rs.pre_move(b, 0, d)
            assert np.array_equal(nb, g['after'][i, d]) and s == g['reward'][i, d] and c == bool(g['changed'][i, d])
        assert rs.game_over(b) == bool(g['game_over'][i])


def test_spawn(golden):
    g = golden('spawn.npz')
    out, tile, pos = rb.spawn_injected(g['boards'], g['r10'], g['k'])
    assert np.array_equal(out, g['after'])
    for i in range(0, len(g['boards']), 29):
        b = g['boards'][i].astype(np.int32)
        rs.spawn_injected(b, int(g['r10'][i]), int(g['k'][i]))
        assert np.array_equal(b, g['after'][i])


@pytest.mark.parametrize('n', [2, 3, 4, 5, 6])
def test_features(golden, n):
    g = golden('features.npz')
    boards = g['boards']
    assert np.array_equal(rb.features(n, boards), g[f'f{n}'])
    for i in range(0, len(boards), 41):
        assert np.array_equal(rs.FEATURES[n](boards[i].astype(np.int64)), g[f'f{n}'][i])


@pytest.mark.parametrize('n', [2, 3, 4, 5, 6])
def test_value_select_update(golden, n):
    g = golden('learner.npz')
    w = formulas.weights(n).astype(np.float64)
    assert np.array_equal(rb.evaluate(n, w, g['boards']), g[f'value{n}'])       # exact: weights are dyadic
    action, value, after, reward, valid, _ = rb.select(n, w, g['sel_boards'])
    assert valid.all()
    assert np.array_equal(action, g[f'action{n}'])
    before = w.copy()
    rb.update(n, w, g['up_states'], g['up_dw'])
    diff = w - before
    nz = np.nonzero(diff)[0]
    assert np.array_equal(nz, g[f'upd_slot{n}'])
    assert np.array_equal(diff[nz], g[f'upd_delta{n}'])


@pytest.mark.parametrize('n', [2, 3, 4])
def test_episode_trace(golden, n):
    """Whole QAgent.episode() of the reference, replayed by the scalar oracle with the same draws."""
    g = golden(f'episode_n{n}.npz')
    draws = [tuple(int(v) for v in row) for row in g['draws']]
    it = iter(draws)

    def feed(n_empty):
        r10, k, ne = next(it)
        assert ne == n_empty
        return r10, k
    w0 = formulas.weights(n, scale=2.0 ** -6)
    agent = rs.Agent(n=n, alpha=float(g['alpha']), weights=w0)
    trace = []
    board, score, moves = agent.episode(feed, trace=trace)
    assert moves == len(g['moves']) - 1 and score == int(g['final_score'])
    assert np.array_equal(board, g['final_board'])
    assert [t['action'] for t in trace] == g['moves'].tolist()
    assert np.array_equal(np.stack([t['board'] for t in trace]), g['boards'])
    dws = np.array([t['dw'] for t in trace if t['dw'] is not None])
    assert np.array_equal(dws, g['rec_dw'])                                     # float64, same op order
    diff = agent.flat_weights() - w0.astype(np.float64)
    nz = np.nonzero(diff)[0]
    assert np.array_equal(nz, g['w_slot'])
    assert np.allclose(diff[nz], g['w_delta'], rtol=0, atol=1e-15)


@pytest.mark.parametrize('n', [2, 4])
def test_batch_td_step_equals_scalar_episode(golden, n):
    """ref_batch.td_step with one lane reproduces the reference episode (the batched semantics reduce to
    online TD(0) at batch 1)."""
    g = golden(f'episode_n{n}.npz')
    draws = iter([tuple(int(v) for v in row) for row in g['draws']])
    w = formulas.weights(n, scale=2.0 ** -6).astype(np.float64)
    w0 = w.copy()
    lanes = rb.Lanes(g['start'][None])
    next(draws), next(draws)                          # the two spawns of Game.__init__ are in `start`

    def feed(idx, n_empty):
        r10, k, ne = next(draws)
        assert ne == n_empty[0]
        return np.array([r10]), np.array([k])
    step = 0
    while not lanes.done[0]:
        assert np.array_equal(lanes.boards[0], g['boards'][step])
        out = rb.td_step(n, w, lanes, float(g['alpha']), feed)
        assert out['action'][0] == g['moves'][step]
        step += 1
    assert lanes.scores[0] == int(g['final_score'])
    diff = w - w0
    nz = np.nonzero(diff)[0]
    assert np.array_equal(nz, g['w_slot'])
    assert np.allclose(diff[nz], g['w_delta'], rtol=0, atol=1e-13)
