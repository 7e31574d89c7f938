"""Parity of the HIP path (through the C ABI, include/g2048.h) against the oracle and the golden vectors.
Integer work (moves, scores, spawns, terminal tests, tuple indices, RNG) must be bit-exact; fp32 values and
weights must be within the tolerance of SURVEY.md §8 a-9: |V_gpu - V_ref| <= 1e-5 * sum|w_i| + 1e-6."""
import importlib

import numpy as np
import pytest

from oracle import ref_batch as rb
from oracle import ref_scalar as rs
from tests import helpers
from tests.golden import formulas

pytestmark = pytest.mark.gpu
pkg = importlib.import_module('2048_amd')
rng_spec = pkg.rng


def Engine(*a, **k):
    return pkg.Engine(*a, **k)


def value_tol(n, weights, boards):
    """1e-5 * sum |w_i| + 1e-6 per board."""
    s = rb.slots(n, boards)
    return 1e-5 * np.abs(np.asarray(weights, np.float64)[s]).sum(axis=1) + 1e-6


# ------------------------------------------------------------------ environment: bit-exact

def test_move_table_all_rows_all_directions(golden):
    g = golden('move_table.npz')
    keys = np.arange(65536)
    line = np.stack([(keys >> 12) & 15, (keys >> 8) & 15, (keys >> 4) & 15, keys & 15], axis=1).astype(np.uint8)
    eng = Engine(65536)
    boards = np.zeros((65536, 4, 4), np.uint8)
    boards[:, 2, :] = line
    eng.set_boards(boards)
    after, reward, changed = eng.move_all()
    assert np.array_equal(after[:, 0, 2, :], g['out'])                       # left
    assert np.array_equal(reward[:, 0], g['score'])
    assert np.array_equal(changed & 1, g['changed'])
    boards = np.zeros((65536, 4, 4), np.uint8)
    boards[:, 1, ::-1] = line
    eng.set_boards(boards)
    after, reward, changed = eng.move_all()
    assert np.array_equal(after[:, 2, 1, ::-1], g['out'])                    # right = left on the mirrored row
    assert np.array_equal(reward[:, 2], g['score'])
    boards = np.zeros((65536, 4, 4), np.uint8)
    boards[:, :, 3] = line
    eng.set_boards(boards)
    after, reward, changed = eng.move_all()
    assert np.array_equal(after[:, 1, :, 3], g['out'])                       # up
    assert np.array_equal((changed >> 1) & 1, g['changed'])
    boards = np.zeros((65536, 4, 4), np.uint8)
    boards[:, ::-1, 0] = line
    eng.set_boards(boards)
    after, reward, changed = eng.move_all()
    assert np.array_equal(after[:, 3, ::-1, 0], g['out'])                    # down
    assert np.array_equal(reward[:, 3], g['score'])
    eng.close()


def test_golden_moves_terminal_spawn(golden):
    g = golden('moves.npz')
    B = len(g['boards'])                                                     # ragged: not a multiple of 64 or 256
    eng = Engine(B)
    eng.set_boards(g['boards'])
    after, reward, changed = eng.move_all()
    assert np.array_equal(after, g['after'])
    assert np.array_equal(reward, g['reward'])
    assert np.array_equal(np.stack([(changed >> d) & 1 for d in range(4)], axis=1), g['changed'])
    over, ne, npairs = eng.terminal()
    assert np.array_equal(over, g['game_over'].astype(bool))
    assert np.array_equal(ne, g['empty_count'])
    assert np.array_equal(npairs, g['adjacent_pair_count'])
    # make_move: board and score
    dirs = (np.arange(B) % 4).astype(np.uint8)
    eng.set_scores(np.arange(B))
    moved = eng.apply_moves(dirs)
    ar = np.arange(B)
    assert np.array_equal(eng.get_boards(), g['after'][ar, dirs])
    assert np.array_equal(eng.get_scores(), ar + g['reward'][ar, dirs])
    assert np.array_equal(moved, g['changed'][ar, dirs].astype(bool))
    eng.close()
    s = golden('spawn.npz')
    eng = Engine(len(s['boards']))
    eng.set_boards(s['boards'])
    eng.spawn_injected(s['r10'], s['k'])
    assert np.array_equal(eng.get_boards(), s['after'])
    eng.close()


def test_full_boards_do_not_spawn_and_rng_untouched():
    full = np.tile(np.array([[1, 2, 1, 2], [2, 1, 2, 1], [1, 2, 1, 2], [2, 1, 2, 1]], np.uint8), (300, 1, 1))
    eng = Engine(300, seed=9)
    eng.set_boards(full)
    st = eng.get_rng()
    r10, k = eng.spawn()
    assert np.array_equal(eng.get_boards(), full) and (k == 255).all() and np.array_equal(eng.get_rng(), st)
    eng.close()


def test_rng_stream_new_games_and_exported_draws():
    B, seed, lane0 = 1000, 2048, 12345
    eng = Engine(B, seed=seed, lane0=lane0)
    state = rng_spec.seed_lanes(seed, lane0, B)
    want = helpers.oracle_new_games(state, np.arange(B))                     # create() starts every lane as a new game
    assert np.array_equal(eng.get_boards(), want)
    assert np.array_equal(eng.get_rng(), state)
    for _ in range(5):                                                       # spawn exports the (r10, k) it used
        before = eng.get_boards()
        r10, k = eng.spawn()
        w10, wk = rng_spec.spawn_draw_np(rng_spec.next_u64_np(state), rb.empty_count(before))
        assert np.array_equal(r10, w10) and np.array_equal(k, wk)
        nb, _, _ = rb.spawn_injected(before, r10, k)                         # replay the draws into the oracle
        assert np.array_equal(eng.get_boards(), nb)
    assert np.array_equal(eng.get_rng(), state)
    eng.reset()
    want = helpers.oracle_new_games(state, np.arange(B))
    assert np.array_equal(eng.get_boards(), want) and (eng.get_scores() == 0).all()
    eng.close()


@pytest.mark.parametrize('B,steps', [(65536, 40), (777, 400)])
def test_step_random_config2(B, steps):
    """BASELINE config 2: env step only, random valid direction — bit-exact boards, scores, RNG and stats."""
    eng = Engine(B, seed=31)
    boards, scores, state = eng.get_boards(), eng.get_scores().astype(np.int64), eng.get_rng()
    eng.step_random(steps)
    want = helpers.oracle_step_random(boards, scores, state, steps)
    assert np.array_equal(eng.get_boards(), boards)
    assert np.array_equal(eng.get_scores(), scores)
    assert np.array_equal(eng.get_rng(), state)
    st = eng.stats()
    assert st['moves'] == want['moves'] and st['episodes'] == want['episodes']
    assert st['score_sum'] == want['score_sum'] and st['best_score'] == want['best']
    assert np.array_equal(np.array(st['max_tile']), want['hist'])
    eng.close()


def test_step_random_without_auto_reset_runs_every_game_to_the_end():
    B = 512
    eng = Engine(B, seed=5)
    eng.set_auto_reset(False)
    boards, scores, state = eng.get_boards(), eng.get_scores().astype(np.int64), eng.get_rng()
    eng.step_random(3000)
    want = helpers.oracle_step_random(boards, scores, state, 3000, auto_reset=False)
    assert want['done'].all()
    assert np.array_equal(eng.get_boards(), boards) and np.array_equal(eng.get_scores(), scores)
    assert rb.game_over(eng.get_boards()).all()
    assert eng.stats()['episodes'] == B
    eng.close()


# ------------------------------------------------------------------ features: bit-exact

@pytest.mark.parametrize('n', [2, 3, 4, 5, 6])
def test_features_golden(golden, n):
    g = golden('features.npz')
    eng = Engine(len(g['boards']), n=n)
    eng.set_boards(g['boards'])
    assert np.array_equal(eng.features(), g[f'f{n}'])
    eng.close()


# ------------------------------------------------------------------ value / select / update

@pytest.mark.parametrize('n', [2, 3, 4, 5, 6])
def test_value_select_update_golden(golden, n):
    g = golden('learner.npz')
    w = formulas.weights(n)
    eng = Engine(len(g['boards']), n=n)
    eng.set_weights(w)
    eng.set_boards(g['boards'])
    v = eng.evaluate()
    assert np.array_equal(v.astype(np.float64), g[f'value{n}'])              # dyadic weights: exact in fp32 too
    eng.close()
    eng = Engine(len(g['sel_boards']), n=n)
    eng.set_weights(w)
    eng.set_boards(g['sel_boards'])
    value, action, v4 = eng.eval_select(want_all=True)
    assert np.array_equal(action, g[f'action{n}'])
    o_action, o_value, _, _, _, o_vals = rb.select(n, w.astype(np.float64), g['sel_boards'])
    assert np.array_equal(v4.astype(np.float64), o_vals) and np.array_equal(value.astype(np.float64), o_value)
    # update: the reference's sparse table difference
    eng.update(g['up_states'], g['up_dw'])
    diff = eng.get_weights().astype(np.float64) - w.astype(np.float64)
    nz = np.nonzero(diff)[0]
    assert np.array_equal(nz, g[f'upd_slot{n}'])
    assert np.allclose(diff[nz], g[f'upd_delta{n}'], rtol=0, atol=1e-6)
    eng.close()


def test_eval_select_config3_full_size():
    """BASELINE config 3: 262 144 boards, n = 3, evaluate + 4-way greedy select, against the oracle."""
    n, B = 3, 262144
    eng = Engine(B, n=n, seed=77)
    eng.step_random(48)                                                      # mid-game-like boards
    w = np.random.RandomState(3).rand(formulas.table_size(n)).astype(np.float32) / 100   # U[0, 0.01) as init_weights
    eng.set_weights(w)
    boards = eng.get_boards()
    value, action, v4 = eng.eval_select(want_all=True)
    o_action, o_value, o_after, _, valid, o_vals = rb.select(n, w.astype(np.float64), boards)
    assert valid.all()
    fin = np.isfinite(o_vals)
    assert np.array_equal(np.isfinite(v4), fin)
    tol = 1e-5 * 52 * 0.01 + 1e-6
    assert np.abs(v4[fin] - o_vals[fin]).max() <= tol
    # the action may only differ where the oracle's two best values are closer than the tolerance
    diff = np.nonzero(action != o_action)[0]
    srt = np.sort(o_vals[diff], axis=1)
    assert (srt[:, -1] - srt[:, -2] <= 2 * tol).all()
    assert len(diff) < B // 1000
    eng.close()


@pytest.mark.parametrize('n', [2, 3, 4, 5, 6])
def test_update_linearity_full_size(n):
    """Size-independent property at 2^20 records: every record adds dw to 8 * num_feat slots, so the table sum
    moves by 8 * F * sum(dw); and update(+dw) followed by update(-dw) restores the table."""
    B = 1 << 20 if n != 6 else 1 << 18
    F = rs.NUM_FEAT[n]
    eng = Engine(B, n=n, seed=4)
    eng.step_random(32)
    states = eng.get_boards()
    dw = ((np.arange(B) % 17) - 8).astype(np.float32) * np.float32(2.0 ** -10)
    eng.update(states, dw)
    w = eng.get_weights()
    assert abs(float(w.astype(np.float64).sum()) - 8 * F * float(dw.astype(np.float64).sum())) < 1e-3 * 8 * F
    # against the oracle on a slice (float64)
    sl = slice(0, 20000)
    ref = np.zeros(formulas.table_size(n), np.float64)
    rb.update(n, ref, states[sl], dw[sl].astype(np.float64))
    eng2 = Engine(20000, n=n)
    eng2.update(states[sl], dw[sl])
    assert np.abs(eng2.get_weights().astype(np.float64) - ref).max() < 1e-4
    eng2.close()
    eng.update(states, -dw)
    assert np.abs(eng.get_weights()).max() < 1e-3                            # dyadic dw: sums cancel up to fp32 order effects
    eng.close()


# ------------------------------------------------------------------ TD(0)

@pytest.mark.parametrize('mode', [1, 0])
@pytest.mark.parametrize('n', [2, 3, 4, 5, 6])
def test_td_single_lane_reproduces_reference_episode(golden, n, mode):
    """Batch 1, same weights, same (r10, k) draws as the reference's QAgent.episode(): every board, action and
    score of the whole game is identical; the learned table matches within fp32 tolerance."""
    g = golden(f'episode_n{n}.npz')
    eng = Engine(1, n=n, seed=int(g['seed']))
    eng.set_auto_reset(False)
    eng.set_update_mode(mode)
    w0 = formulas.weights(n, scale=2.0 ** -6)
    eng.set_weights(w0)
    assert np.array_equal(eng.get_boards()[0], g['start'])                   # same two opening spawns
    steps = len(g['moves']) - 1
    for t in range(steps):
        assert np.array_equal(eng.get_boards()[0], g['boards'][t]), f'board differs at move {t}'
        assert eng.get_scores()[0] == g['scores'][t]
        eng.td_steps(float(g['alpha']), 1)
    assert np.array_equal(eng.get_boards()[0], g['final_board'])
    assert eng.get_scores()[0] == int(g['final_score'])
    _, _, flags = eng.get_carry()
    assert flags[0] & 2                                                      # DONE
    st = eng.stats()
    assert st['episodes'] == 1 and st['moves'] == steps and st['score_sum'] == int(g['final_score'])
    diff = eng.get_weights().astype(np.float64) - w0.astype(np.float64)
    want = np.zeros_like(diff)
    want[g['w_slot']] = g['w_delta']
    assert np.abs(diff - want).max() < 5e-5      # fp32 table, |dw| up to ~10, up to 16 adds per slot
    eng.td_steps(float(g['alpha']), 3)                                       # a finished lane stays put
    assert np.array_equal(eng.get_boards()[0], g['final_board'])
    eng.close()


@pytest.mark.parametrize('mode', [1, 0])
@pytest.mark.parametrize('n', [2, 3, 4, 5, 6])
def test_td_steps_batch_vs_oracle(n, mode):
    """Synchronous batched TD(0), 4096 lanes: every step is checked against the float64 oracle started from the
    device's own state (dyadic weights, so the comparison of boards, scores, RNG and labels is exact)."""
    B = 4096
    alpha = formulas.exact_alpha(n)
    eng = Engine(B, n=n, seed=100 + n)
    eng.set_auto_reset(False)
    eng.set_update_mode(mode)
    eng.step_random(30)                                                      # mid-game boards
    for t in range(5):
        helpers.check_td_step(eng, n, alpha, formulas.weights(n, scale=2.0 ** -(4 + t)))
    eng.close()


@pytest.mark.parametrize('rule', ['sum', 'mean'])
@pytest.mark.parametrize('B,n', [(3, 5), (1000, 4), (70001, 5), (1237, 6), (777, 3)])
def test_td_ragged_lane_counts(B, n, rule):
    """Lane counts that are not a multiple of the wave, the workgroup or the record-scan stride."""
    eng = Engine(B, n=n, seed=300 + B)
    eng.set_auto_reset(False)
    eng.set_update_rule(1 if rule == 'mean' else 0)
    eng.step_random(25)
    for t in range(3):
        helpers.check_td_step(eng, n, formulas.exact_alpha(n), formulas.weights(n, scale=2.0 ** -(5 + t)), rule=rule)
    eng.close()


@pytest.mark.parametrize('n', [2, 3, 4, 5, 6])
def test_td_mean_rule_vs_oracle(n):
    """The optional per-slot mean rule (g2048_set_update_rule(1)) against its restatement in the oracle."""
    B = 4096
    eng = Engine(B, n=n, seed=200 + n)
    eng.set_auto_reset(False)
    eng.set_update_rule(1)
    eng.step_random(30)
    for t in range(4):
        helpers.check_td_step(eng, n, formulas.exact_alpha(n), formulas.weights(n, scale=2.0 ** -(4 + t)), rule='mean')
    eng.close()


@pytest.mark.parametrize('poll', [1, 0])
def test_td_large_batch_vs_oracle_through_replans(poll, monkeypatch):
    """131 072 lanes, n = 5: the size at which the XCD-resident plan and the measured-cost replans (every 8 steps) are
    in use; every step is still checked against the float64 oracle.  Both ways the planner's statistics reach the host: the
    sequence word the apply kernel stores behind them (default) and the event behind the apply kernel (G2048_PLAN_POLL=0);
    a burst of steps without an accessor in between, so that the host runs ahead of the device, sits in the middle."""
    monkeypatch.setenv('G2048_PLAN_POLL', str(poll))
    n, B = 5, 1 << 17
    eng = Engine(B, n=n, seed=77)
    eng.set_auto_reset(False)
    eng.step_random(40)
    for t in range(18):
        if t == 9:
            eng.set_weights(formulas.weights(n, scale=2.0 ** -6))
            eng.td_steps(formulas.exact_alpha(n) * 2.0 ** -12, 24)
        helpers.check_td_step(eng, n, formulas.exact_alpha(n), formulas.weights(n, scale=2.0 ** -(4 + t % 4)))
    plan = eng.debug_owner_plan()
    assert len(plan) % 8 == 0                                                # parts come in multiples of the 8 XCDs
    eng.close()


@pytest.mark.parametrize('n,B', [(2, 3000), (3, 3001), (4, 4096), (5, 5000), (6, 1237)])
def test_td_hot_set_path_vs_oracle(n, B, monkeypatch):
    """k_td_play's LDS hot set (the first 2 048 entries of every four-cell table in memory order — small tiles — read from
    an LDS copy; the other lanes' gathers issued under their exec mask through an asm load; on by default from 2^18 lanes)
    forced on at a small batch: mid-game boards mix hot and cold tuples, random boards of tiles 0..13 are almost all cold; every
    step is checked against the float64 oracle as everywhere else.  n = 2, 3: their LDS forms (the whole table; the entries
    whose three cells are all below 8, the others through the same asm loads), which otherwise start at 2^14 / 2^17 lanes."""
    monkeypatch.setenv('G2048_PLAY_HOT_MIN', '1')
    monkeypatch.setenv('G2048_PLAY_HOT', '2')                # (2: for n = 6 too, where it is off by default)
    eng = Engine(B, n=n, seed=900 + n)
    eng.set_auto_reset(False)
    eng.step_random(30)
    for t in range(3):
        helpers.check_td_step(eng, n, formulas.exact_alpha(n), formulas.weights(n, scale=2.0 ** -(4 + t)))
    r = np.random.RandomState(n)
    boards = (r.randint(0, 14, (B, 4, 4)) * (r.rand(B, 4, 4) < 0.8)).astype(np.uint8)
    boards[:64] = r.randint(0, 6, (64, 4, 4))                # some all-hot boards too
    boards[:, 0, 0] = 0                                      # an empty cell and a tile: every lane has a move
    boards[:, 1, 1] = np.maximum(boards[:, 1, 1], 1)
    eng.set_boards(boards)
    for t in range(2):
        helpers.check_td_step(eng, n, formulas.exact_alpha(n), formulas.weights(n, scale=2.0 ** -(5 + t)))
    eng.close()


@pytest.mark.parametrize('rule', ['sum', 'mean'])
@pytest.mark.parametrize('order', ['auto', 'hot', 'plain'])
@pytest.mark.parametrize('n', [4, 5])
def test_td_four_cell_orbit_table_orders(n, order, rule, monkeypatch):
    """The four-cell orbits' accumulation tables (features.hpp, quad_place; g2048.hip, QuadOrder): hot-first (every index whose
    cells are all <= 10 in chunk 0) for young boards, the plain index order once bigger tiles send more than a few per cent of the
    adds elsewhere, back to hot-first at g2048_reset.  Every step is the oracle's step under either order, pinned
    (G2048_QUAD_ORDER) or chosen by the context, through the switch and back; which order is in force shows in the plan:
    chunks 0, 5, 10, 15, 20 (the orbits' first chunks) have workgroups only under the hot-first order."""
    if order != 'auto':
        monkeypatch.setenv('G2048_QUAD_ORDER', order)
    monkeypatch.setenv('G2048_REPLAN_EVERY', '1')            # the planner looks at the measured shares after every step
    B = 8192
    eng = Engine(B, n=n, seed=4100 + n)
    eng.set_auto_reset(False)
    eng.set_update_rule(1 if rule == 'mean' else 0)
    eng.step_random(30)

    def first_chunks_in_plan():
        plan = eng.debug_owner_plan().astype(np.int64)
        return sorted(set(plan[:, 1].tolist()) & {0, 5, 10, 15, 20})

    for t in range(3):
        helpers.check_td_step(eng, n, formulas.exact_alpha(n), formulas.weights(n, scale=2.0 ** -(4 + t)), rule=rule)
    assert first_chunks_in_plan() == ([] if order == 'plain' else [0, 5, 10, 15, 20])
    r = np.random.RandomState(n)
    boards = (r.randint(0, 14, (B, 4, 4)) * (r.rand(B, 4, 4) < 0.8)).astype(np.uint8)       # most tuples hold a cell >= 11
    boards[:512] = r.randint(0, 9, (512, 4, 4))
    boards[:, 0, 0] = 0
    boards[:, 1, 1] = np.maximum(boards[:, 1, 1], 1)
    eng.set_boards(boards)
    for t in range(4):
        helpers.check_td_step(eng, n, formulas.exact_alpha(n), formulas.weights(n, scale=2.0 ** -(5 + t % 3)), rule=rule)
    assert first_chunks_in_plan() == ([0, 5, 10, 15, 20] if order == 'hot' else [])          # (auto: switched)
    eng.reset()
    eng.step_random(20)
    for t in range(3):
        helpers.check_td_step(eng, n, formulas.exact_alpha(n), formulas.weights(n, scale=2.0 ** -(4 + t)), rule=rule)
    assert first_chunks_in_plan() == ([] if order == 'plain' else [0, 5, 10, 15, 20])          # (auto: hot-first again)
    eng.close()


@pytest.mark.parametrize('lag', [2, 0])
@pytest.mark.parametrize('n', [3, 4, 5, 6])
def test_lane_sort_is_invisible(n, lag, monkeypatch):
    """g2048_set_lane_sort: lanes re-ordered by their big-tile pattern every few steps (k_td_play reads them through the
    sorted permutation).  Every step is still the oracle's step, lane for lane (the accessors restore the identity
    order), including lanes that finish (auto-reset off: DONE lanes travel with the others).  lag 2 (the default): the
    sort runs on a side stream and is applied two steps after the boards it looked at — an accessor in between drops it;
    lag 0: sorted in line."""
    monkeypatch.setenv('G2048_SORT_MIN', '1')
    monkeypatch.setenv('G2048_SORT_LAG', str(lag))
    monkeypatch.setenv('G2048_PLAY_HOT_MIN', '1' if lag else str(1 << 30))      # with and without the LDS hot set (the kernel that re-orders has both forms)
    monkeypatch.setenv('G2048_PLAY_HOT', '2')
    B = 3000
    eng = Engine(B, n=n, seed=600 + n)
    eng.set_auto_reset(False)
    eng.set_lane_sort(2)
    eng.step_random(80)                                      # late enough for big tiles and for some games to be over
    for t in range(5):
        helpers.check_td_step(eng, n, formulas.exact_alpha(n), formulas.weights(n, scale=2.0 ** -(4 + t)))
    eng.td_steps(formulas.exact_alpha(n) * 2.0 ** -12, 9)    # re-orderings on top of each other, no accessor in between
    # (a small alpha: the table of the last check stays in place here, and 3000 lanes at the per-game alpha would blow it up)
    helpers.check_td_step(eng, n, formulas.exact_alpha(n), formulas.weights(n, scale=2.0 ** -5))
    eng.close()


@pytest.mark.parametrize('by_value', [1, 0])
@pytest.mark.parametrize('B', [1, 777, 4096, 4097, (1 << 18) + 13])
def test_lane_order_is_a_sorted_permutation(B, by_value, monkeypatch):
    """The hand-written lane re-order (k_sort_count / k_sort_scan / k_sort_scatter): its output is a permutation of the
    positions, non-decreasing in the key.  The key (default): value >> 1 of every tile above 32, by cell, folded into 16 bits
    by a multiplicative hash — lanes with the same big tiles in the same places share a bucket; G2048_SORT_VALUES=0: one bit per
    cell — tile above 32 — row 0 in the top nibble (round 2's key).  Ragged sizes around the 4 096-lane tile of a workgroup;
    run twice (the bucket counters must be back at zero)."""
    monkeypatch.setenv('G2048_SORT_VALUES', str(by_value))
    eng = Engine(B, n=4, seed=31 + B)
    eng.step_random(150)                                     # mid-game boards: many distinct patterns
    boards = eng.get_boards().reshape(B, 16)
    want = np.zeros(B, np.uint32)
    if by_value:
        h = np.zeros(B, np.uint64)
        for cell in range(16):                               # row-major, as the kernel walks the board
            v = boards[:, cell].astype(np.uint64)
            h = (h * np.uint64(0x9E3779B1) + np.where(v > 5, (v >> np.uint64(1)) & np.uint64(7), 0).astype(np.uint64)) & np.uint64(0xFFFFFFFF)
        want = ((h >> np.uint64(16)) ^ (h & np.uint64(0xFFFF))).astype(np.uint32)
    else:
        for cell in range(16):
            r, c = divmod(cell, 4)
            want |= (boards[:, cell] > 5).astype(np.uint32) << (4 * (3 - r) + c)
    for _ in range(2):
        perm, keys = eng.debug_lane_order()
        assert np.array_equal(keys, want.astype(np.uint16))
        assert np.array_equal(np.sort(perm), np.arange(B, dtype=np.uint32))
        assert np.all(np.diff(keys[perm].astype(np.int64)) >= 0)
    assert len(np.unique(keys)) > 1 or B == 1
    eng.close()


def test_lane_sort_same_games_at_scale():
    """Same seed with and without lane re-ordering, auto-reset on, 2^17 lanes, n = 4, 40 steps.  A lane's game does not
    depend on where the lane sits, so both runs play the same games — up to the run-to-run noise any two runs have (the
    workgroups' partial sums meet in float atomics, whose order is not fixed: the tables agree to fp32 rounding, and a
    greedy choice can flip where two values are an ulp apart).  The recorded games of the watched lanes follow the
    lanes, not the positions: each replays to its recorded score."""
    n, B, steps = 4, 1 << 17, 40
    out = []
    for every in (0, 3):
        eng = Engine(B, n=n, seed=12)
        eng.init_weights(seed=3, scale=0.01)
        eng.log_enable(256, 2048)
        eng.set_lane_sort(every)
        eng.td_steps(0.25 * 17 / (8 * B), steps)
        out.append((eng.get_boards(), eng.get_scores(), eng.get_rng(), eng.get_weights(), eng.stats(), eng.last_move(), eng.log_meta()))
        if every:
            eng.td_steps(0.25 * 17 / (8 * B), 5)             # (and keeps going after the accessors restored the order)
            assert eng.stats()['moves'] == B * (steps + 5)
            eng.set_auto_reset(False)
            eng.td_steps(0.25 * 17 / (8 * B), 400)           # let games end: finished lanes travel with the others
            meta = eng.log_meta()
            import game2048.r_learning as rl
            agent = rl.QAgent(name='t', storage='local', console='local', n=n, with_weights=False)
            done = 0
            for lane in range(256):
                for slot in (0, 1):
                    length, score, flags = int(meta[lane, 3 + 2 * slot]), int(meta[lane, 4 + 2 * slot]), int(meta[lane, 7])
                    if length and not (flags >> slot) & 5 and done < 12:
                        agent._game_from_log(eng, lane, slot, length, score, verify=True)      # asserts that the record replays to its end
                        done += 1
            assert done >= 8
        eng.close()
    a, b = out
    same = (a[0].reshape(B, 16) == b[0].reshape(B, 16)).all(axis=1)
    # (two runs WITHOUT re-ordering differ in 400-1200 of these lanes after 40 steps, tools/sortdbg.py)
    assert same.mean() > 0.98, f'{(~same).sum()} of {B} lanes hold different boards'
    assert np.array_equal(a[1][same], b[1][same]) and np.array_equal(a[2][same], b[2][same]) and np.array_equal(a[5][same], b[5][same])
    assert np.abs(a[3] - b[3]).max() < 0.2 and np.abs(a[3] - b[3]).mean() < 1e-4
    assert a[4]['moves'] == b[4]['moves'] == B * steps and abs(a[4]['episodes'] - b[4]['episodes']) <= 64


@pytest.mark.parametrize('n', [2, 3, 5, 6])
def test_td_config4_full_size_owner_path(n):
    """BASELINE config 4 at its full size — 2^20 lanes, n = 5, the LDS-owner update the bench times — and the per-GPU
    workload of config 5 (n = 6: the binned f_6 update on top of it).  Per step:
    a slice of lanes is replayed by the float64 oracle (their choices depend only on the table before the step: boards,
    scores, RNG, carried state and labels bit for bit), every live lane moves once, and the table's total change equals
    8 F sum(dw) with the records' dw rebuilt from the lanes' exported state (r_learning.py:240,248).
    n = 2, 3 (round 3): the same at full size through their own kernels — the table (n = 2) / its small-tile part (n = 3) in LDS
    for the gathers (k_td_play_lds2 / _lds3) and the orbit-reduced owner update (24 / 52 adds per record instead of 192 / 416)."""
    B, F = 1 << 20, {2: 24, 3: 52, 5: 21, 6: 33}[n]
    alpha = formulas.exact_alpha(n)
    eng = Engine(B, n=n, seed=4)
    eng.set_auto_reset(False)
    eng.step_random(60)
    lo, hi = 500000, 500000 + 8192
    for t in range(3):
        w = formulas.weights(n, scale=2.0 ** -(5 + t))
        eng.set_weights(w)
        boards0, scores0 = eng.get_boards(), eng.get_scores()
        prev0, label0, flags0 = eng.get_carry()
        rng0 = eng.get_rng()
        lanes = rb.Lanes(boards0[lo:hi], scores0[lo:hi])
        lanes.prev, lanes.label = prev0[lo:hi].copy(), label0[lo:hi].astype(np.float64)
        lanes.has_prev, lanes.done = (flags0[lo:hi] & 1).astype(bool), (flags0[lo:hi] & 2).astype(bool)
        draws = helpers.SpecDraws(rng0[lo:hi].copy())
        rb.td_step(n, w.astype(np.float64).copy(), lanes, alpha, draws)
        moves0 = eng.stats()['moves']
        eng.td_steps(alpha, 1)
        boards1, scores1 = eng.get_boards(), eng.get_scores()
        prev1, label1, flags1 = eng.get_carry()
        live = (flags0 & 2) == 0
        assert np.array_equal(boards1[lo:hi], lanes.boards) and np.array_equal(scores1[lo:hi], lanes.scores)
        assert np.array_equal(eng.get_rng()[lo:hi], draws.state)
        sl_live = live[lo:hi]
        assert np.array_equal(prev1[lo:hi][sl_live], lanes.prev[sl_live])
        assert np.array_equal(label1[lo:hi].astype(np.float64), lanes.label)
        assert np.array_equal((flags1[lo:hi] & 2).astype(bool), lanes.done)
        assert eng.stats()['moves'] - moves0 == int(live.sum())              # every live lane made one move
        # sum of the step's dw from the exported lane state (dyadic weights: labels are exact)
        had = live & ((flags0 & 1) != 0)
        dw1 = (scores1.astype(np.float64) - scores0 + label1.astype(np.float64) - label0.astype(np.float64)) * alpha / F
        ended = live & ((flags1 & 2) != 0)
        dw2 = -label1.astype(np.float64) * alpha / F
        total = 8.0 * F * (dw1[had].sum() + dw2[ended].sum())
        mass = 8.0 * F * (np.abs(dw1[had]).sum() + np.abs(dw2[ended]).sum())
        got = (eng.get_weights().astype(np.float64) - w.astype(np.float64)).sum()
        assert abs(got - total) <= 1e-6 * mass + 1e-6, (got, total)
    assert not rb.game_over(eng.get_boards()[(eng.get_carry()[2] & 2) == 0]).any()    # no dead board stays live
    # ---- bursts with NO accessor in between (round 4): an accessor restores the lane order and drops a pending re-order, so the
    # step-by-step part above never runs the kernels the bench times between its barriers — k_td_play_hot<5, 512, PERM = true>,
    # steps on re-ordered lanes, a replan mid-burst.  17 steps = one sort period (8) + its lag (2) + seven steps on permuted lanes.
    K = 17
    w = formulas.weights(n, scale=2.0 ** -6)
    eng.set_weights(w)
    boards0, scores0, rng0 = eng.get_boards(), eng.get_scores(), eng.get_rng()
    prev0, label0, flags0 = eng.get_carry()
    # (a) alpha = 0: no record changes the table, so the float64 oracle can replay the slice through all K steps
    lanes = rb.Lanes(boards0[lo:hi], scores0[lo:hi])
    lanes.prev, lanes.label = prev0[lo:hi].copy(), label0[lo:hi].astype(np.float64)
    lanes.has_prev, lanes.done = (flags0[lo:hi] & 1).astype(bool), (flags0[lo:hi] & 2).astype(bool)
    draws = helpers.SpecDraws(rng0[lo:hi].copy())
    w64 = w.astype(np.float64)
    for _ in range(K):
        rb.td_step(n, w64, lanes, 0.0, draws)
    moves0 = eng.stats()['moves']
    eng.td_steps(0.0, K)
    boards1, scores1 = eng.get_boards(), eng.get_scores()
    prev1, label1, flags1 = eng.get_carry()
    assert np.array_equal(boards1[lo:hi], lanes.boards) and np.array_equal(scores1[lo:hi], lanes.scores)
    assert np.array_equal(eng.get_rng()[lo:hi], draws.state)
    alive = ~lanes.done
    assert np.array_equal(prev1[lo:hi][alive], lanes.prev[alive]) and np.array_equal(label1[lo:hi].astype(np.float64), lanes.label)
    assert np.array_equal((flags1[lo:hi] & 2).astype(bool), lanes.done)
    assert np.array_equal(eng.get_weights(), w)                               # alpha = 0: the table did not move
    assert 0 < eng.stats()['moves'] - moves0 <= K * int(((flags0 & 2) == 0).sum())
    # (b) a small alpha: the records of a lane telescope — sum_t dw1_t = (score_K - score_0 + label_K - label_0) alpha / F for a
    # lane that carried a state into the burst, and a game that ends inside it closes with -label_K alpha / F (r_learning.py:240,248)
    # — so the table's total change over the burst is known from the lane state before and after, whatever order the lanes were in
    small = alpha * 2.0 ** -12
    boards0, scores0 = boards1, scores1
    label0, flags0 = label1, flags1
    eng.td_steps(small, K)
    scores1 = eng.get_scores()
    _, label1, flags1 = eng.get_carry()
    live0 = (flags0 & 2) == 0
    had = live0 & ((flags0 & 1) != 0)
    ended = live0 & ((flags1 & 2) != 0)
    gain = (scores1.astype(np.float64) - scores0) + label1.astype(np.float64)
    per_lane = np.where(had, gain - label0.astype(np.float64), 0.0)
    # (a lane without a carried state makes no record on its first move; its later records telescope from that move's value,
    # which the lane state does not show: at this point of the test every live lane has a state)
    assert not (live0 & ~had).any()
    per_lane = per_lane - np.where(ended, label1.astype(np.float64), 0.0)
    total = 8.0 * F * per_lane[live0].sum() * small / F
    mass = 8.0 * small * (np.abs(scores1.astype(np.float64) - scores0) + np.abs(label1) + np.abs(label0))[live0].sum()
    got = (eng.get_weights().astype(np.float64) - w64).sum()
    assert abs(got - total) <= 2e-5 * mass + 1e-6, (got, total, mass)
    assert not rb.game_over(eng.get_boards()[(eng.get_carry()[2] & 2) == 0]).any()
    eng.close()


@pytest.mark.parametrize('n', [2, 4, 6])
def test_lookahead_steps_at_depth_zero_are_the_greedy_steps(n):
    """g2048_lookahead_steps (csrc/lookahead.hip: roots -> tree -> k_la_pick) with depth 0 is Game.trial_run's greedy loop, and so is
    g2048_td_steps with alpha = 0 (k_td_play): two contexts with the same lanes must play the same games — boards, scores, RNG
    streams, statistics, what every lane did last — through two independent sets of kernels.  And a deeper search still only makes
    legal moves: every live lane moves once per step."""
    B = 3000
    a, b = Engine(B, n=n, seed=77), Engine(B, n=n, seed=77)
    w = formulas.weights(n, scale=2.0 ** -5)
    for eng in (a, b):
        eng.set_weights(w)
        eng.set_auto_reset(False)
    for _ in range(3):
        a.td_steps(0.0, 25)
        b.lookahead_steps(0, 1, 0, 0, 25)
        assert np.array_equal(a.get_boards(), b.get_boards()) and np.array_equal(a.get_scores(), b.get_scores())
        assert np.array_equal(a.get_rng(), b.get_rng()) and np.array_equal(a.last_move(), b.last_move())
        sa, sb = a.stats(), b.stats()
        assert (sa['moves'], sa['episodes'], sa['score_sum'], sa['valid_dirs'], sa['max_tile']) == (sb['moves'], sb['episodes'], sb['score_sum'], sb['valid_dirs'], sb['max_tile'])
    assert np.array_equal(a.get_weights(), w) and np.array_equal(b.get_weights(), w)      # nothing was learned
    live = int(((b.get_carry()[2] & 2) == 0).sum())
    moves0 = b.stats()['moves']
    b.lookahead_steps(2, 2, 8, 0, 1)
    assert b.stats()['moves'] - moves0 == live
    # a tile limit ends games that are not over (game_logic.py:177): with limit 2^5 every lane stops at once
    c = Engine(64, n=n, seed=3)
    c.set_weights(w)
    c.set_auto_reset(False)
    c.lookahead_steps(1, 2, 16, 5, 400)
    assert c.stats()['episodes'] == 64 and (c.get_boards().reshape(64, 16).max(axis=1) >= 5).sum() >= 60
    for eng in (a, b, c):
        eng.close()


@pytest.mark.parametrize('n', [3, 5, 6])
def test_td_rule_and_mode_switches_mid_run(n):
    """Switching between the sum and the mean rule, and between the two update kernels, between steps: the orbit tables
    are double-buffered and the count buffers are only maintained under the mean rule, so every switch must start from
    clean accumulators (odd numbers of steps in between flip the buffer parity)."""
    B = 4096
    eng = Engine(B, n=n, seed=400 + n)
    eng.set_auto_reset(False)
    eng.step_random(30)
    k = 0

    def steps(count, rule):
        nonlocal k
        for _ in range(count):
            helpers.check_td_step(eng, n, formulas.exact_alpha(n), formulas.weights(n, scale=2.0 ** -(4 + k % 5)), rule=rule)
            k += 1

    eng.set_update_rule(1); steps(3, 'mean')
    eng.set_update_rule(0); steps(3, 'sum')
    eng.set_update_rule(1); steps(2, 'mean')
    eng.set_update_rule(0); steps(1, 'sum')
    eng.set_update_mode(0); steps(2, 'sum')                                  # global-atomics kernel
    eng.set_update_mode(1); steps(2, 'sum')
    eng.set_update_rule(1); steps(1, 'mean')
    eng.close()


@pytest.mark.parametrize('mode', [1, 0])
def test_td_batch_until_all_games_end(mode):
    """256 lanes played to the end with learning on: the per-step check holds through terminal updates and DONE."""
    n, B = 2, 256
    eng = Engine(B, n=n, seed=99)
    eng.set_auto_reset(False)
    eng.set_update_mode(mode)
    w = formulas.weights(n, scale=2.0 ** -5)
    ended = 0
    for t in range(1500):
        _, out = helpers.check_td_step(eng, n, formulas.exact_alpha(n), w)
        ended += int(out['over'].sum())
        if ended == B:
            break
    assert ended == B and eng.stats()['episodes'] == B
    eng.close()


@pytest.mark.parametrize('mode', [0, 1])
def test_td_whole_game_fp32_model_bit_exact(mode):
    """One lane, one whole game with random (non-dyadic) weights, against the oracle run in the device's
    arithmetic (float32, same operation order): boards and scores are bit-identical for the whole game, and so
    is every weight (up to the order of the last step's two records on a shared slot)."""
    n, alpha, seed = 2, 0.25, 321
    eng = Engine(1, n=n, seed=seed)
    eng.set_auto_reset(False)
    eng.set_update_mode(mode)
    w0 = (np.random.RandomState(1).rand(formulas.table_size(n)) / 100).astype(np.float32)
    eng.set_weights(w0)
    w = w0.copy()
    lanes = rb.Lanes(eng.get_boards())
    draws = helpers.SpecDraws(eng.get_rng())
    steps = 0
    while not lanes.done[0] and steps < 5000:
        rb.td_step(n, w, lanes, alpha, draws)
        steps += 1
    assert lanes.done[0]
    eng.td_steps(alpha, steps)
    assert np.array_equal(eng.get_boards(), lanes.boards)
    assert eng.get_scores()[0] == lanes.scores[0]
    got = eng.get_weights()
    if mode == 0:       # one fp32 add per record and slot, as the model: only slots shared by the last step's two
        differ = np.nonzero(got != w)[0]        # records may differ (their add order is free)
        assert len(differ) <= 8 * 24
    # mode 1 sums a step's adds to a slot in LDS first and adds the sum once: same value up to fp32 rounding
    assert np.abs(got.astype(np.float64) - w).max() <= 1e-5 * max(1.0, float(np.abs(w).max()))
    eng.close()


@pytest.mark.parametrize('n', [2, 3, 4, 5, 6])
def test_update_modes_agree_at_scale(n):
    """The LDS-owner update (mode 1) and the global-atomics update (mode 0) add the same records: after the same
    steps from the same state the two tables agree within fp32 accumulation tolerance, at a batch large enough
    to use the multi-part plan (2^17 lanes)."""
    B = 1 << 17
    tables = []
    for mode in (1, 0):
        eng = Engine(B, n=n, seed=55)
        eng.set_update_mode(mode)
        eng.step_random(40)
        eng.set_weights(formulas.weights(n, scale=2.0 ** -4))
        eng.td_steps(0.0, 1)                                                 # step 1 only sets `state` (alpha 0: no records)
        eng.td_steps(2.0 ** -12, 1)                                          # step 2: same choices in both modes, then the adds
        tables.append((eng.get_weights().astype(np.float64), eng.get_boards(), eng.stats()['moves']))
        eng.close()
    (w1, b1, m1), (w0, b0, m0) = tables
    assert np.array_equal(b1, b0) and m1 == m0 == 42 * B                    # 40 random + 2 TD steps per lane
    base = formulas.weights(n, scale=2.0 ** -4).astype(np.float64)
    moved = np.abs(w0 - base)
    assert moved.max() > 0
    # the hottest slot takes up to 8 * B adds; summed one by one in fp32 (mode 0) the rounding noise is about
    # eps * |sum| * sqrt(adds), the LDS-owner path sums hierarchically and is more accurate: allow 4x that
    assert np.abs(w1 - w0).max() <= 2.0 ** -23 * moved.max() * 4.0 * np.sqrt(8.0 * B)


def test_td_auto_reset_and_stats_consistency():
    """Full-size property: with auto-reset on, every lane makes one move per step; finished games are counted
    and restarted; the table stays finite."""
    n, B, steps = 4, 1 << 16, 600
    eng = Engine(B, n=n, seed=8)
    eng.init_weights(seed=1, scale=0.01)
    eng.td_steps(0.25 * 17 / (8 * B), steps)                                  # batch rule: alpha * F / (8 * lanes)
    st = eng.stats()
    assert st['moves'] == B * steps
    assert st['episodes'] > 0 and st['score_sum'] > 0 and st['best_score'] >= st['score_sum'] / st['episodes']
    assert sum(st['max_tile']) == st['episodes']
    boards = eng.get_boards()
    assert not rb.game_over(boards).any()                                    # finished lanes were restarted at once
    assert np.isfinite(eng.get_weights()).all()
    eng.close()


def test_weight_delta_protocol_on_device():
    """g2048_delta_begin / extract / apply (the per-epoch all-reduce plumbing): D = W - W0 is exactly what the epoch's
    steps added, and W0 + k*D rebuilds the table a k-rank sum all-reduce would produce."""
    import ctypes
    n, B = 4, 8192
    eng = Engine(B, n=n, seed=21)
    w0 = formulas.weights(n, scale=2.0 ** -5)
    eng.set_weights(w0)
    eng.delta_begin()
    eng.td_steps(2.0 ** -16, 3)
    w1 = eng.get_weights()
    assert np.abs(w1 - w0).max() > 0
    eng.delta_extract()                                                      # into the context's own buffer
    eng.delta_apply()                                                        # one rank: W = W0 + D = W
    assert np.allclose(eng.get_weights(), w1, rtol=1e-6, atol=1e-7)      # fp32: W0 + (W - W0)
    # a second epoch measures from the new snapshot
    eng.td_steps(2.0 ** -16, 2)
    w2 = eng.get_weights()
    eng.delta_extract()
    eng.delta_apply()
    assert np.allclose(eng.get_weights(), w2, rtol=1e-6, atol=1e-7)
    ptr, count = eng.weights_ptr()
    assert ptr and count == formulas.table_size(n) and eng.delta_ptr()
    eng.timer_start()
    eng.td_steps(0.0, 2)
    assert eng.timer_stop() > 0
    eng.close()


def test_shared_table_contexts_and_last_move():
    """g2048_create_shared: a second set of lanes over the same table sees the first one's updates;
    g2048_get_last_move reports direction / new tile / end of game of every lane."""
    n = 3
    main = Engine(2048, n=n, seed=4)
    main.set_weights(formulas.weights(n, scale=2.0 ** -6))
    solo = Engine(1, seed=99, lane0=1 << 40, share_table_of=main)
    solo.set_auto_reset(False)
    probe = np.array([[1, 2, 3, 4], [0, 5, 1, 0], [2, 2, 11, 15], [14, 13, 0, 1]], np.uint8)
    solo.set_boards(probe[None])
    v0 = solo.evaluate()[0]
    main.update(probe[None], [0.5])
    solo.set_boards(probe[None])
    assert abs(solo.evaluate()[0] - (v0 + 0.5 * 52)) < 1e-3                  # identity image adds dw to each of the 52 slots
    solo.reset()
    before = solo.get_boards()[0].copy()
    solo.td_steps(0.1, 1)
    lm = int(solo.last_move()[0])
    assert lm & 4 and lm & (1 << 10)
    after, reward, changed = rb.move_all(before[None])
    d = lm & 3
    assert changed[0, d]
    cell, tile = (lm >> 4) & 15, (lm >> 8) & 3
    want = after[0, d].copy()
    want[cell >> 2, cell & 3] = tile
    assert np.array_equal(solo.get_boards()[0], want) and solo.get_scores()[0] == reward[0, d]
    solo.close()
    main.close()


# ------------------------------------------------------------------ error behaviour

def test_owner_plan_diagnostics_cover_every_chunk_once():
    """g2048_debug_owner_plan: the LDS-owner plan in use and the clocks of its last launch.  The parts of every chunk
    must tile the records exactly once, and every workgroup must have run (end clock after start clock)."""
    n, B = 5, 1 << 17
    eng = Engine(B, n=n, seed=12)
    eng.init_weights(seed=3, scale=0.01)
    eng.td_steps(0.25 * 21 / (8.0 * B), 20)                                  # two replans with measured costs
    eng.sync()
    plan = eng.debug_owner_plan().astype(np.int64)
    assert len(plan) > 0 and len(plan) <= 1024
    assert (plan[:, 5] > plan[:, 4]).all()
    for chunk in np.unique(plan[:, 1]):
        rows = plan[plan[:, 1] == chunk]
        nparts = rows[0, 3]
        assert (rows[:, 3] == nparts).all() and len(np.unique(rows[:, 0])) == 1
        assert sorted(rows[:, 2].tolist()) == list(range(nparts))
    eng.close()


def test_error_codes():
    from ctypes import byref, c_void_p
    lib = pkg.load_library()
    ctx = c_void_p()
    assert lib.g2048_create(0, 16, 7, 0, 0, byref(ctx)) == -1                # bad n-tuple
    assert lib.g2048_create(0, 0, 2, 0, 0, byref(ctx)) == -1                 # empty batch
    assert lib.g2048_create(99, 16, 2, 0, 0, byref(ctx)) == -1               # no such device
    eng = Engine(16)                                                         # environment only
    with pytest.raises(Exception) as e:
        eng.features()
    assert e.value.status == -4
    eng.close()
    eng = Engine(16, n=2)
    assert eng.lib.g2048_weights_set(eng.ctx, None, 5) == -1                 # null buffer
    five = np.zeros(5, np.float32)
    assert eng.lib.g2048_weights_set(eng.ctx, five.ctypes.data, 5) == -1     # wrong count
    assert b'count' in eng.lib.g2048_last_error(eng.ctx)
    eng.close()
