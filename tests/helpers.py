"""Shared test helpers: oracle-side replays of the device's batched loops, driven by the RNG spec."""
import importlib

import numpy as np

from oracle import ref_batch as rb

rng_spec = importlib.import_module('2048_amd.rng')


def kth_set_bit(mask, k):
    """position of the k-th set bit of each 4-bit mask (vectorised)."""
    out = np.zeros(len(mask), np.int64)
    seen = np.zeros(len(mask), np.int64)
    found = np.zeros(len(mask), bool)
    for d in range(4):
        bit = (mask >> d) & 1
        hit = (bit == 1) & (seen == k) & ~found
        out[hit] = d
        found |= hit
        seen += bit
    assert found.all()
    return out


def oracle_new_games(state, idx):
    """Game.__init__ for lanes idx (two spawns from each lane's stream); returns boards [len(idx),4,4]."""
    boards = np.zeros((len(idx), 4, 4), np.uint8)
    sub = state[idx]
    for _ in range(2):
        r10, k = rng_spec.spawn_draw_np(rng_spec.next_u64_np(sub), rb.empty_count(boards))
        boards, _, _ = rb.spawn_injected(boards, r10, k)
    state[idx] = sub
    return boards.astype(np.uint8)


def oracle_step_random(boards, scores, state, nsteps, auto_reset=True):
    """Oracle replay of g2048_step_random: uniformly random valid direction, move, spawn, terminal check.
    Mutates boards/scores/state; returns dict(episodes, moves, score_sum, best, hist, done)."""
    B = len(boards)
    done = np.zeros(B, bool)
    stats = dict(episodes=0, moves=0, score_sum=0, best=0, hist=np.zeros(20, np.int64))
    for _ in range(nsteps):
        idx = np.nonzero(~done)[0]
        if not len(idx):
            break
        after, reward, changed = rb.move_all(boards[idx])
        mask = (changed * (1 << np.arange(4))).sum(axis=1)
        assert (mask != 0).all(), 'live lanes always have a move in these tests'
        sub = state[idx]
        j = rng_spec.pick_draw_np(rng_spec.next_u64_np(sub), changed.sum(axis=1))
        d = kth_set_bit(mask, j.astype(np.int64))
        ar = np.arange(len(idx))
        nb = after[ar, d]
        scores[idx] += reward[ar, d]
        stats['moves'] += len(idx)
        r10, k = rng_spec.spawn_draw_np(rng_spec.next_u64_np(sub), rb.empty_count(nb))
        nb, _, _ = rb.spawn_injected(nb, r10, k)
        state[idx] = sub
        boards[idx] = nb
        over = rb.game_over(nb)
        fin = idx[over]
        if len(fin):
            stats['episodes'] += len(fin)
            stats['score_sum'] += int(scores[fin].sum())
            stats['best'] = max(stats['best'], int(scores[fin].max()))
            np.add.at(stats['hist'], boards[fin].reshape(len(fin), 16).max(axis=1), 1)
            if auto_reset:
                boards[fin] = oracle_new_games(state, fin)
                scores[fin] = 0
            else:
                done[fin] = True
    stats['done'] = done
    return stats


class SpecDraws:
    """draws(idx, n_empty) callback for rb.td_step that follows each lane's own xoroshiro stream."""

    def __init__(self, state):
        self.state = state

    def __call__(self, idx, n_empty):
        sub = self.state[idx]
        u = rng_spec.next_u64_np(sub)
        self.state[idx] = sub
        return rng_spec.spawn_draw_np(u, n_empty)


def lanes_from_engine(eng):
    """oracle Lanes mirroring the device's lane state (boards, scores, `state`, `old_label`, flags)."""
    lanes = rb.Lanes(eng.get_boards(), eng.get_scores())
    prev, label, flags = eng.get_carry()
    lanes.prev = prev.copy()
    lanes.label = label.astype(np.float64)
    lanes.has_prev = (flags & 1).astype(bool)
    lanes.done = (flags & 2).astype(bool)
    return lanes


def check_td_step(eng, n, alpha, weights, rule='sum'):
    """One synchronous TD(0) step on the device against the float64 oracle started from the device's own lane
    state.  `weights` must be dyadic (tests/golden/formulas.weights): value sums are then exact in fp32 and
    float64 alike, so the greedy choices agree exactly and boards / scores / RNG / carry are compared bit for
    bit; the table is compared within fp32 accumulation tolerance.  Returns the max weight error."""
    eng.set_weights(weights)
    lanes = lanes_from_engine(eng)
    live = ~lanes.done
    w = weights.astype(np.float64)
    draws = SpecDraws(eng.get_rng())
    out = rb.td_step(n, w, lanes, alpha, draws, rule)
    eng.td_steps(alpha, 1)
    assert np.array_equal(eng.get_boards(), lanes.boards), 'boards differ from the oracle'
    assert np.array_equal(eng.get_scores(), lanes.scores), 'scores differ'
    assert np.array_equal(eng.get_rng(), draws.state), 'RNG state differs'
    prev, label, flags = eng.get_carry()
    assert np.array_equal(prev[live], lanes.prev[live]), 'carried afterstate differs'
    assert np.array_equal(label.astype(np.float64), lanes.label), 'old_label differs (dyadic weights: must be exact)'
    assert np.array_equal((flags & 2).astype(bool), lanes.done)
    # fp32 atomic accumulation: a slot that receives `count` adds of total magnitude `mass` may be off by about
    # eps * mass * sqrt(count) (each add rounds at the running sum's ulp); allow 4x that plus 4 ulp.
    got = eng.get_weights().astype(np.float64)
    if rule == 'mean':                    # S / C with both summed in fp32: a relative error of a few ulp of the largest |dw|
        err = np.abs(got - w)
        bound = 1e-5 * (np.abs(out['rec_dw']).max() if len(out['rec_dw']) else 0.0) + 1e-7 * np.abs(w) + 1e-9
        bad = np.nonzero(err > bound)[0]
        assert len(bad) == 0, f'{len(bad)} slots off under the mean rule, worst {err[bad].max()} at slot {bad[err[bad].argmax()]}'
        return float(err.max()), out
    mass = np.abs(weights.astype(np.float64))
    count = np.zeros_like(mass)
    rb.update(n, mass, out['rec_states'], np.abs(out['rec_dw']))
    rb.update(n, count, out['rec_states'], np.ones(len(out['rec_dw'])))
    tol = 2.0 ** -23 * mass * (4.0 * np.sqrt(count) + 4.0) + 1e-9
    err = np.abs(got - w)
    bad = np.nonzero(err > tol)[0]
    assert len(bad) == 0, f'{len(bad)} slots outside fp32 accumulation tolerance, worst {err[bad].max()} at slot {bad[err[bad].argmax()]}'
    return float(err.max()), out
