"""Deterministic inputs shared by the golden generator (make_golden.py) and the tests.

Nothing here comes from the reference: these are formulas for *inputs* (boards, weights, dw)
that are too large to store (the n=6 table is 382 MB) or convenient to regenerate.
"""
import numpy as np

NUM_FEAT = {2: 24, 3: 52, 4: 17, 5: 21, 6: 33}
GROUPS = {
    2: [(24, 16 ** 2)],
    3: [(52, 16 ** 3)],
    4: [(17, 16 ** 4)],
    5: [(17, 16 ** 4), (4, 16 ** 5)],
    6: [(17, 16 ** 4), (4, 16 ** 5), (12, 14 ** 6)],
}


def feature_sizes(n):
    out = []
    for count, size in GROUPS[n]:
        out += [size] * count
    return out


def table_size(n):
    return int(sum(feature_sizes(n)))


def weights(n, scale=1.0):
    """float32-exact pseudo-random weights in [-0.5, 0.5) * scale, multiples of 2^-10 * scale
    (scale must be a power of two): sums of <= 33 of them are exact in float32 and float64 alike."""
    total = table_size(n)
    with np.errstate(over='ignore'):
        slot = np.arange(total, dtype=np.uint64)
        h = (slot * np.uint64(0x9E3779B97F4A7C15) + np.uint64(0x1234567)) >> np.uint64(40)
        h = (h & np.uint64(0x3FF)).astype(np.int64) - 512
    return (h.astype(np.float32) * np.float32(2.0 ** -10 * scale))


def update_dws(count):
    """dw values that are multiples of 2^-12 in (-1, 1)."""
    k = (np.arange(count, dtype=np.int64) * 2654435761 % 8191) - 4095
    return k.astype(np.float64) * 2.0 ** -12


def exact_alpha(n):
    """alpha such that alpha / num_feat is a power of two (2^-7 .. 2^-6)."""
    return {2: 24 / 128, 3: 52 / 256, 4: 17 / 64, 5: 21 / 128, 6: 33 / 128}[n]


EDGE_BOARDS = [
    [[0] * 4] * 4,
    [[1, 2, 1, 2], [2, 1, 2, 1], [1, 2, 1, 2], [2, 1, 2, 1]],          # full, no move: game over
    [[1, 1, 1, 1]] * 4,
    [[1, 1, 2, 2], [0, 0, 0, 0], [3, 3, 3, 0], [4, 0, 4, 4]],
    [[2, 0, 2, 2], [1, 1, 2, 0], [0, 0, 0, 1], [1, 0, 0, 1]],
    [[15, 15, 14, 14], [13, 13, 13, 13], [15, 0, 0, 14], [14, 15, 14, 15]],
    [[1, 2, 3, 4], [0, 5, 1, 0], [2, 2, 11, 15], [14, 13, 0, 1]],
    [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10, 11], [12, 13, 14, 15]],
    [[15, 14, 13, 12], [11, 10, 9, 8], [7, 6, 5, 4], [3, 2, 1, 0]],
    [[1, 2, 3, 4], [2, 3, 4, 5], [3, 4, 5, 6], [4, 5, 6, 0]],            # one empty corner
    [[1, 2, 3, 4], [2, 3, 4, 5], [3, 4, 5, 6], [4, 5, 6, 6]],            # full, exactly one merge
    [[0, 0, 0, 0], [0, 0, 0, 0], [0, 0, 0, 0], [0, 0, 0, 1]],
    [[1, 0, 0, 0], [0, 0, 0, 0], [0, 0, 0, 0], [0, 0, 0, 0]],
    [[13, 13, 14, 14], [13, 14, 13, 14], [12, 13, 14, 15], [15, 14, 13, 12]],
    [[3, 3, 3, 3], [3, 3, 3, 3], [3, 3, 3, 3], [3, 3, 3, 3]],
    [[1, 0, 1, 0], [0, 1, 0, 1], [1, 0, 1, 0], [0, 1, 0, 1]],
]


def fixture_boards(played=None):
    """~4096 boards uint8[N,4,4]: hand-made edge cases, random fills at several densities and tile
    ranges (incl. 14/15 for the base-14 clamp of f_6), merge-heavy small alphabets, and optionally
    boards from actual play (passed in by the generator)."""
    r = np.random.RandomState(20481)
    out = [np.array(b, np.uint8) for b in EDGE_BOARDS]
    for b in list(out):                                   # all 8 symmetries of the edge boards
        for kk in range(1, 4):
            out.append(np.rot90(b, kk).copy())
        out.append(b.T.copy())
    for density in (0.2, 0.5, 0.8, 1.0):
        for top in (2, 3, 5, 8, 11, 13, 15):
            for _ in range(80):
                tiles = r.randint(1, top + 1, (4, 4))
                mask = r.rand(4, 4) < density
                out.append((tiles * mask).astype(np.uint8))
    for _ in range(600):                                  # merge-heavy
        alphabet = r.choice([0, 1, 2, 3, 14, 15], size=r.randint(2, 4), replace=False)
        out.append(r.choice(alphabet, size=(4, 4)).astype(np.uint8))
    if played is not None:
        out += [np.array(b, np.uint8) for b in played]
    return np.stack(out)


def lookahead_draws(board, k):
    """The sampled chance nodes of Game.look_forward (game_logic.py:221-225) for the fixtures: which k empty cells get a
    tile and which tile (2 with probability 0.1), as a function of the BOARD only — so that the reference's depth-first
    walk and the batched level-by-level walk draw the same tiles whatever their visiting order.
    Returns (cells flat index [k], tiles [k])."""
    b = np.ascontiguousarray(board, np.uint8)
    r = np.random.RandomState(int.from_bytes(b.tobytes()[:8], 'little') % (2 ** 31))
    empties = np.nonzero(b.reshape(16) == 0)[0]
    cells = r.permutation(empties)[:k]
    tiles = np.where(r.rand(k) < 0.1, 2, 1)
    return cells, tiles
