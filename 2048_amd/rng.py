"""Host mirror of the device RNG (spec, not a fallback).

The reference draws with Python's Mersenne Twister (`random.randrange(10)` then
`random.choice(empties)`, game2048/game_logic.py:112-116).  The device cannot reproduce
MT19937, so "identical RNG draws" is defined at the level of the (r10, k) draw pair:

    u   = xoroshiro128++ next()                       (one 64-bit draw per spawn)
    r10 = (hi32(u) * 10)      >> 32                   tile = 2 iff r10 == 0 else 1
    k   = (lo32(u) * n_empty) >> 32                   k-th empty cell, row-major

Lane state is seeded from splitmix64(seed + global_lane).  This module is pure-int Python
(and a numpy-vectorised twin) so that the golden-fixture generator can feed exactly these
draws into the reference, and the tests can check the device stream bit for bit.
"""
import numpy as np

MASK64 = (1 << 64) - 1
GOLDEN = 0x9E3779B97F4A7C15


def splitmix64(x):
    """One splitmix64 step: returns (new_state, output)."""
    x = (x + GOLDEN) & MASK64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return x, z ^ (z >> 31)


def seed_lane(seed, lane):
    """(s0, s1) for one lane."""
    x = (seed + lane) & MASK64
    x, s0 = splitmix64(x)
    x, s1 = splitmix64(x)
    if (s0 | s1) == 0:
        s0 = 1
    return s0, s1


def _rotl(x, k):
    return ((x << k) | (x >> (64 - k))) & MASK64


def next_u64(s0, s1):
    """xoroshiro128++: returns (u, s0', s1')."""
    u = (_rotl((s0 + s1) & MASK64, 17) + s0) & MASK64
    s1 ^= s0
    s0n = _rotl(s0, 49) ^ s1 ^ ((s1 << 21) & MASK64)
    s1n = _rotl(s1, 28)
    return u, s0n, s1n


def spawn_draw(u, n_empty):
    """(r10, k) from one 64-bit draw."""
    return ((u >> 32) * 10) >> 32, ((u & 0xFFFFFFFF) * n_empty) >> 32


def pick_draw(u, n_valid):
    """index of the chosen valid direction for the random policy."""
    return ((u >> 32) * n_valid) >> 32


class LaneRng:
    """Scalar stream for one lane (used by the fixture generator's `random` shim)."""

    def __init__(self, seed, lane=0, state=None):
        self.s0, self.s1 = state if state is not None else seed_lane(seed, lane)

    def next(self):
        u, self.s0, self.s1 = next_u64(self.s0, self.s1)
        return u


# ---------------------------------------------------------------- chance nodes of the look-ahead (g2048_boards_look_forward)

def lookahead_rng(board16, salt=(0, 0)):
    """The stream the look-ahead samples a chance node from (Game.look_forward, game_logic.py:221-225: `random.sample` of the
    empty cells, then `randrange(10)` per tile).  The reference draws depth-first from one global Mersenne Twister; the device
    walks all trees level by level, so the draws are defined as a FUNCTION of the node — a xoroshiro128++ stream keyed by the
    node's board and a salt (for a game: the lane's RNG state when the move is chosen, so that the same position gets other
    samples in another game or at another move) — and are the same in whatever order the nodes are visited:

        lo, hi = the 16 board bytes as two little-endian u64
        x = salt0 ^ lo;            a  = splitmix64(x)        (x advances by the golden constant, as in seed_lane)
        x = (x ^ hi) + salt1;      s0 = splitmix64(x)
        x = x ^ a;                 s1 = splitmix64(x)
    """
    b = bytes(bytearray(int(v) for v in board16))
    assert len(b) == 16
    lo, hi = int.from_bytes(b[:8], 'little'), int.from_bytes(b[8:], 'little')
    x = (salt[0] ^ lo) & MASK64
    x, a = splitmix64(x)
    x = ((x ^ hi) + salt[1]) & MASK64
    x, s0 = splitmix64(x)
    x ^= a
    x, s1 = splitmix64(x)
    if (s0 | s1) == 0:
        s0 = 1
    return LaneRng(0, state=(s0, s1))


def lookahead_draws(board16, k, salt=(0, 0)):
    """[(cell, tile)] x k: the sampled chance nodes of `board16` — k distinct empty cells (flat row-major index) without
    replacement, one draw each: (r10, j) = spawn_draw(u, cells still free) picks the j-th free cell in row-major order and the
    tile (2 iff r10 == 0), exactly as a spawn does."""
    g = lookahead_rng(board16, salt)
    free = [i for i, v in enumerate(board16) if int(v) == 0]
    out = []
    for _ in range(k):
        r10, j = spawn_draw(g.next(), len(free))
        out.append((free.pop(j), 2 if r10 == 0 else 1))
    return out


# ---------------------------------------------------------------- numpy twin (vectorised)

def seed_lanes(seed, lane0, count):
    """uint64[count, 2] lane states for global lanes lane0 .. lane0+count-1."""
    with np.errstate(over='ignore'):
        x = (np.uint64(seed & MASK64) + np.arange(lane0, lane0 + count, dtype=np.uint64))
        out = np.empty((count, 2), dtype=np.uint64)
        for j in range(2):
            x = x + np.uint64(GOLDEN)
            z = x.copy()
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            out[:, j] = z ^ (z >> np.uint64(31))
        dead = (out[:, 0] | out[:, 1]) == 0
        out[dead, 0] = 1
    return out


def next_u64_np(state):
    """Advance uint64[B,2] in place, return uint64[B] outputs."""
    def rotl(v, k):
        return (v << np.uint64(k)) | (v >> np.uint64(64 - k))
    with np.errstate(over='ignore'):
        s0 = state[:, 0].copy()
        s1 = state[:, 1].copy()
        u = rotl(s0 + s1, 17) + s0
        s1 ^= s0
        state[:, 0] = rotl(s0, 49) ^ s1 ^ (s1 << np.uint64(21))
        state[:, 1] = rotl(s1, 28)
    return u


def spawn_draw_np(u, n_empty):
    hi = (u >> np.uint64(32)).astype(np.uint64)
    lo = (u & np.uint64(0xFFFFFFFF)).astype(np.uint64)
    r10 = (hi * np.uint64(10)) >> np.uint64(32)
    k = (lo * n_empty.astype(np.uint64)) >> np.uint64(32)
    return r10.astype(np.uint8), k.astype(np.uint8)


def pick_draw_np(u, n_valid):
    hi = (u >> np.uint64(32)).astype(np.uint64)
    return ((hi * n_valid.astype(np.uint64)) >> np.uint64(32)).astype(np.uint8)


# ---------------------------------------------------------------- weight initialisation (g2048_weights_init)

def init_weights_np(count, seed, scale=0.01, first=0):
    """The table k_weights_init builds — init_weights (r_learning.py:136-149) is U[0, 0.01) per slot; here slot i gets
    float32(top 24 bits of splitmix64 output of state seed + i) * 2^-24 * scale, counter-based so that every rank
    builds the same table.  Returns float32[count] for slots first .. first + count - 1."""
    with np.errstate(over='ignore'):
        x = np.uint64(seed & MASK64) + np.arange(first, first + count, dtype=np.uint64) + np.uint64(GOLDEN)
        z = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24) * np.float32(scale)
