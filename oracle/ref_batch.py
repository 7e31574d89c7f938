"""ORACLE (test infrastructure, not product code) — NumPy batch restatement of the reference path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Same algorithm as oracle/ref_scalar.py (which keeps the reference's one-board structure), but
vectorised over a leading batch axis so parity tests at 2^16..2^20 boards finish in seconds.
Boards are uint8/int arrays [B, 4, 4] of log2 tiles (0 = empty), row-major as in
game2048/game_logic.py:62.

Parity status: PINNED — checked against golden vectors produced by importing the reference in
the build container (tests/golden/make_golden.py) and against ref_scalar on random boards.
"""
import numpy as np

from . import ref_scalar as rs

# ------------------------------------------------------------------ row table as arrays (a-1)

_LUT = None


def row_lut():
    """(out[65536,4] u8, score[65536] i64, changed[65536] bool), key = a<<12|b<<8|c<<4|d with a the
    leftmost cell — array form of Game.table (game_logic.py:18-39)."""
    global _LUT
    if _LUT is None:
        out = np.zeros((65536, 4), dtype=np.uint8)
        score = np.zeros(65536, dtype=np.int64)
        changed = np.zeros(65536, dtype=bool)
        for key in range(65536):
            line = ((key >> 12) & 15, (key >> 8) & 15, (key >> 4) & 15, key & 15)
            o, s, c = rs.slide_row_left(line)
            out[key], score[key], changed[key] = o, s, c
        _LUT = (out, score, changed)
    return _LUT


def as_boards(b):
    b = np.asarray(b)
    if b.ndim == 2 and b.shape[1] == 16:
        b = b.reshape(-1, 4, 4)
    assert b.ndim == 3 and b.shape[1:] == (4, 4)
    return b


# ------------------------------------------------------------------ environment (a-2 .. a-5)

def move_left(boards):
    """Game._left over a batch (game_logic.py:123-134).  Tiles must be in [0, 15] (table domain)."""
    out, score, changed = row_lut()
    b = boards.astype(np.int64)
    if b.size and b.max() > 15:
        raise ValueError('tile > 15 is outside the reference move table (game_logic.py:20-23)')
    key = (b[..., 0] << 12) | (b[..., 1] << 8) | (b[..., 2] << 4) | b[..., 3]        # [B,4]
    after = out[key]                                                                  # [B,4,4]
    gained = (score[key] * changed[key]).sum(axis=1)
    return after, gained, changed[key].any(axis=1)


def move(boards, direction):
    """Game.pre_move over a batch (game_logic.py:136-142): rot90 by `direction`, left, rot90 back."""
    boards = as_boards(boards)
    view = np.rot90(boards, direction, axes=(1, 2)) if direction else boards
    after, gained, moved = move_left(view)
    if direction:
        after = np.rot90(after, 4 - direction, axes=(1, 2))
    return np.ascontiguousarray(after), gained, moved


def move_all(boards):
    """All four directions: after[B,4,4,4] u8, reward[B,4] i64, changed[B,4] bool."""
    boards = as_boards(boards)
    B = boards.shape[0]
    after = np.empty((B, 4, 4, 4), dtype=np.uint8)
    reward = np.empty((B, 4), dtype=np.int64)
    changed = np.empty((B, 4), dtype=bool)
    for d in range(4):
        after[:, d], reward[:, d], changed[:, d] = move(boards, d)
    return after, reward, changed


def empty_count(boards):
    """game_logic.py:101-103."""
    boards = as_boards(boards)
    return 16 - np.count_nonzero(boards.reshape(len(boards), 16), axis=1)


def adjacent_pair_count(boards):
    """game_logic.py:105-107."""
    boards = as_boards(boards)
    h = (boards[:, :, :3] == boards[:, :, 1:]).reshape(len(boards), -1).sum(axis=1)
    v = (boards[:, :3, :] == boards[:, 1:, :]).reshape(len(boards), -1).sum(axis=1)
    return h + v


def game_over(boards):
    """game_logic.py:109-110."""
    return (empty_count(boards) == 0) & (adjacent_pair_count(boards) == 0)


def spawn_injected(boards, r10, k):
    """create_new_tile/new_tile (game_logic.py:112-121) with injected draws, over a batch.
    tile = 2 iff r10 == 0; placed at the k-th empty cell in row-major order.  Returns a new array
    plus (tile, flat position)."""
    boards = as_boards(boards)
    flat = boards.reshape(len(boards), 16).copy()
    empties = flat == 0
    rank = np.cumsum(empties, axis=1) - 1                      # index of each empty cell in the list
    hit = empties & (rank == np.asarray(k).reshape(-1, 1))
    assert (hit.sum(axis=1) == 1).all(), 'k must be < n_empty'
    pos = hit.argmax(axis=1)
    tile = np.where(np.asarray(r10) == 0, 2, 1).astype(flat.dtype)
    flat[np.arange(len(flat)), pos] = tile
    return flat.reshape(-1, 4, 4), tile, pos


# ------------------------------------------------------------------ n-tuple encoders (a-6 .. a-8)

def features(n, boards):
    """f_n over a batch (r_learning.py:17-69): int64[B, num_feat].  Slicing restated on axes 1, 2."""
    x = as_boards(boards).astype(np.int64)
    B = len(x)

    def flat(a):
        return a.reshape(B, int(np.prod(a.shape[1:])))

    if n == 2:
        return np.concatenate([flat(x[:, 0:3, :] * 16 + x[:, 1:4, :]),
                               flat(x[:, :, 0:3] * 16 + x[:, :, 1:4])], axis=1)
    if n == 3:
        def enc(a, b, c):
            return flat(a * 256 + b * 16 + c)
        tl, tr = x[:, 0:3, 0:3], x[:, 0:3, 1:4]
        bl, br = x[:, 1:4, 0:3], x[:, 1:4, 1:4]
        return np.concatenate([enc(x[:, 0:2, :], x[:, 1:3, :], x[:, 2:4, :]),
                               enc(x[:, :, 0:2], x[:, :, 1:3], x[:, :, 2:4]),
                               enc(bl, br, tr), enc(tl, bl, br), enc(tl, tr, br), enc(tl, bl, tr)], axis=1)
    parts = [x[:, 0, :] * 4096 + x[:, 1, :] * 256 + x[:, 2, :] * 16 + x[:, 3, :],
             x[:, :, 0] * 4096 + x[:, :, 1] * 256 + x[:, :, 2] * 16 + x[:, :, 3],
             flat(x[:, 0:3, 0:3] * 4096 + x[:, 1:4, 0:3] * 256 + x[:, 0:3, 1:4] * 16 + x[:, 1:4, 1:4])]
    if n >= 5:
        parts.append(flat(x[:, 1:3, 1:3] * 65536 + x[:, 0:2, 1:3] * 4096 + x[:, 1:3, 0:2] * 256
                          + x[:, 2:4, 1:3] * 16 + x[:, 1:3, 2:4]))
    if n == 6:
        y = np.minimum(x, 13)
        m = [14 ** 5, 14 ** 4, 14 ** 3, 14 ** 2, 14, 1]
        parts.append(flat(m[0] * y[:, 0:2, 0:3] + m[1] * y[:, 1:3, 0:3] + m[2] * y[:, 2:4, 0:3]
                          + m[3] * y[:, 0:2, 1:4] + m[4] * y[:, 1:3, 1:4] + m[5] * y[:, 2:4, 1:4]))
        parts.append(flat(m[0] * y[:, 0:3, 0:2] + m[1] * y[:, 0:3, 1:3] + m[2] * y[:, 0:3, 2:4]
                          + m[3] * y[:, 1:4, 0:2] + m[4] * y[:, 1:4, 1:3] + m[5] * y[:, 1:4, 2:4]))
    if n not in (4, 5, 6):
        raise ValueError(n)
    return np.concatenate(parts, axis=1)


def slots(n, boards):
    """Flat weight-table slot of every feature: offset[i] + f_n(board)[i]."""
    offs, _ = rs.feature_offsets(n)
    return features(n, boards) + offs[None, :]


def d4_images(boards):
    """The 8 images QAgent.update visits (r_learning.py:207-214), batch form, same order."""
    cur = as_boards(boards)
    out = []
    for _ in range(4):
        out.append(cur)
        t = np.transpose(cur, (0, 2, 1))
        out.append(t)
        cur = np.rot90(cur, 1, axes=(1, 2))
    return out


# ------------------------------------------------------------------ value / select / update (a-9 .. a-12)

def evaluate(n, weights, boards):
    """QAgent.evaluate (r_learning.py:202-203): left-to-right sum of one weight per feature.
    Arithmetic follows the dtype of `weights`: float64 is the reference; float32 is the arithmetic model of the
    device (same IEEE adds in the same order), used where a test wants bit-equal trajectories."""
    s = slots(n, boards)
    w = np.asarray(weights)
    assert w.dtype in (np.float64, np.float32)
    total = np.zeros(len(s), dtype=w.dtype)
    for i in range(s.shape[1]):
        total = total + w[s[:, i]]
    return total


def select(n, weights, boards):
    """Greedy afterstate choice (r_learning.py:229-237): first max over changed directions, start at -inf.
    Returns action u8, value f64, afterstate [B,4,4], reward i64, any_valid bool, all values [B,4]."""
    after, reward, changed = move_all(boards)
    B = len(after)
    vals = np.full((B, 4), -np.inf)
    for d in range(4):
        idx = np.nonzero(changed[:, d])[0]
        if len(idx):
            vals[idx, d] = evaluate(n, weights, after[idx, d])
    action = np.argmax(vals, axis=1)                       # argmax returns the first maximum
    ar = np.arange(B)
    return (action.astype(np.uint8), vals[ar, action], after[ar, action], reward[ar, action],
            changed.any(axis=1), vals)


def update(n, weights, states, dw, rule='sum'):
    """QAgent.update (r_learning.py:207-214) for a batch of (state, dw) records: weights[slot] += dw for
    every feature of the 8 symmetric images (coincident slots accumulate).  In place on float64 `weights`.
    rule='mean' is NOT the reference: it restates the device's optional per-slot mean rule (a slot targeted by C of
    the batch's dw, summing to S, moves by S / C; records with dw == 0 do not count), for its parity test."""
    dw = np.asarray(dw, dtype=weights.dtype)
    if rule == 'sum':
        for img in d4_images(states):
            s = slots(n, img)
            np.add.at(weights, s.ravel(), np.repeat(dw, s.shape[1]))
        return weights
    total = np.zeros_like(weights)
    count = np.zeros_like(weights)
    live = dw != 0
    for img in d4_images(states):
        s = slots(n, img)[live]
        np.add.at(total, s.ravel(), np.repeat(dw[live], s.shape[1]))
        np.add.at(count, s.ravel(), 1.0)
    hit = count > 0
    weights[hit] += total[hit] / count[hit]
    return weights


# ------------------------------------------------------------------ batched synchronous TD(0) step (a-11, a-13)

class Lanes:
    """Per-lane episode state carried between steps (the locals of QAgent.episode, r_learning.py:224-252)."""

    def __init__(self, boards, scores=None):
        self.boards = as_boards(boards).astype(np.uint8).copy()
        B = len(self.boards)
        self.scores = np.zeros(B, dtype=np.int64) if scores is None else np.asarray(scores, dtype=np.int64).copy()
        self.prev = np.zeros((B, 4, 4), dtype=np.uint8)        # `state`
        self.label = np.zeros(B, dtype=np.float64)             # `old_label`
        self.has_prev = np.zeros(B, dtype=bool)                # `state is not None`
        self.done = np.zeros(B, dtype=bool)                    # episode finished (no auto reset)
        self.moves = np.zeros(B, dtype=np.int64)


def td_step(n, weights, lanes, alpha, draws, rule='sum'):
    """One synchronous board-step for every live lane: the body of the while loop of QAgent.episode
    (r_learning.py:228-246) plus, for lanes whose game ends after the spawn, the terminal update
    (r_learning.py:247-249).  All lanes read the same `weights`; every (state, dw) record of the step is
    then applied with `update` (a sum, so lane order is irrelevant).  With one lane this is exactly the
    reference's online TD.  `draws(lane_idx, n_empty) -> (r10, k)` supplies spawn draws.  Arithmetic follows
    weights.dtype (float64 = reference, float32 = device model; see `evaluate`).
    Returns dict of per-lane arrays for inspection."""
    dt = weights.dtype.type
    F = dt(rs.NUM_FEAT[n])
    alpha = dt(alpha)
    live = ~lanes.done
    idx = np.nonzero(live)[0]
    action, value, after, reward, any_valid, vals = select(n, weights, lanes.boards[idx])
    assert any_valid.all(), 'a live lane must have a move'
    value = value.astype(weights.dtype)               # (select keeps values in a float64 array)
    rec_states, rec_dw = [], []
    hp = lanes.has_prev[idx]
    dw1 = (reward.astype(weights.dtype) + value - lanes.label[idx].astype(weights.dtype)) * alpha / F   # r_learning.py:240
    rec_states.append(lanes.prev[idx][hp])
    rec_dw.append(dw1[hp])
    # commit afterstate, r_learning.py:242-245
    lanes.scores[idx] += reward
    lanes.moves[idx] += 1
    lanes.prev[idx] = after
    lanes.label[idx] = value
    lanes.has_prev[idx] = True
    # spawn, r_learning.py:246
    r10, k = draws(idx, empty_count(after))
    spawned, _, _ = spawn_injected(after, r10, k)
    lanes.boards[idx] = spawned
    over = game_over(spawned)
    dw2 = -value * alpha / F                                         # r_learning.py:248
    rec_states.append(after[over])
    rec_dw.append(dw2[over])
    lanes.done[idx[over]] = True
    rec_states, rec_dw = np.concatenate(rec_states), np.concatenate(rec_dw)
    update(n, weights, rec_states, rec_dw, rule)
    return dict(rec_states=rec_states, rec_dw=rec_dw, lanes=idx, action=action, value=value, values4=vals, reward=reward, dw=np.where(hp, dw1, 0.0),
                over=over, dw_term=np.where(over, dw2, 0.0))
