"""ORACLE (test infrastructure, not product code) — scalar CPU restatement of the reference path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
It restates, one board at a time and with the reference's own data structures (65 536-entry
row dictionary, np.rot90 for directions, NumPy slicing for the n-tuple encoders, nested
Python lists of float64 weights, per-move update over 8 symmetries), the algorithm of
/root/reference/game2048/game_logic.py and r_learning.py.  Because it keeps the reference's
structure it is also what bench.py times as the single-core `cpu_baseline` ("port").

Parity status: PINNED — checked against golden vectors produced by importing the reference
itself in the build container (tests/golden/make_golden.py, tests/test_oracle_golden.py).
"""
import numpy as np


# ------------------------------------------------------------------ row table (a-1)

def slide_row_left(line):
    """One row moved left.  Restates create_table, game_logic.py:18-39.

    Non-zero tiles are packed to the left, then one left-to-right pass merges equal
    neighbours (x, x) -> (x+1, gap) adding 2**(x+1) to the score, then gaps are closed.
    A tile produced by a merge is never merged again in the same move.
    Returns (tuple4, score, changed).
    """
    tiles = [t for t in line if t != 0]
    gained = 0
    i = 0
    while i + 1 < len(tiles):
        if tiles[i] != 0 and tiles[i] == tiles[i + 1]:
            tiles[i] += 1
            gained += 1 << tiles[i]
            tiles[i + 1] = 0
        i += 1
    tiles = [t for t in tiles if t != 0]
    out = tuple(tiles + [0] * (4 - len(tiles)))
    return out, gained, out != tuple(line)


def build_row_table():
    """dict (a,b,c,d) -> (row4, score, changed) over [0,16)^4 — Game.table, game_logic.py:51."""
    table = {}
    for key in np.ndindex(16, 16, 16, 16):
        table[key] = slide_row_left(key)
    return table


_ROW_TABLE = None


def row_table():
    global _ROW_TABLE
    if _ROW_TABLE is None:
        _ROW_TABLE = build_row_table()
    return _ROW_TABLE


# ------------------------------------------------------------------ environment (a-2 .. a-5)

ACTIONS = {0: 'left', 1: 'up', 2: 'right', 3: 'down'}      # game_logic.py:50


def move_left(board, score):
    """Game._left, game_logic.py:123-134: every row through the table; score only from changed rows."""
    table = row_table()
    out = board.copy()
    moved = False
    total = score
    for r in range(4):
        new_line, gain, changed = table[tuple(int(v) for v in board[r])]
        if changed:
            moved = True
            total += gain
            out[r] = new_line
    return out, total, moved


def pre_move(board, score, direction):
    """Game.pre_move, game_logic.py:136-142: rotate by `direction` quarter turns, move left, rotate back."""
    view = np.rot90(board, direction) if direction else board
    out, total, moved = move_left(view, score)
    if direction:
        out = np.rot90(out, 4 - direction)
    return out, total, moved


def empty_cells(board):
    """Game.empty, game_logic.py:96-99: empty cells in row-major order."""
    rr, cc = np.where(board == 0)
    return list(zip(rr.tolist(), cc.tolist()))


def empty_count(board):
    """game_logic.py:101-103."""
    return 16 - int(np.count_nonzero(board))


def adjacent_pair_count(board):
    """game_logic.py:105-107: number of equal horizontal + vertical neighbour pairs."""
    horiz = int(np.count_nonzero(board[:, :3] == board[:, 1:]))
    vert = int(np.count_nonzero(board[:3, :] == board[1:, :]))
    return horiz + vert


def game_over(board):
    """game_logic.py:109-110."""
    return empty_count(board) == 0 and adjacent_pair_count(board) == 0


def spawn_injected(board, r10, k):
    """create_new_tile + new_tile, game_logic.py:112-121, with the two draws injected:
    r10 plays random.randrange(10) (tile 2 iff 0), k indexes the row-major empty list
    (random.choice).  Mutates `board`; returns (tile, (r, c))."""
    cells = empty_cells(board)
    tile = 1 if r10 else 2
    pos = cells[k]
    board[pos] = tile
    return tile, pos


def new_game(draws):
    """Game.__init__, game_logic.py:55-66: zero board + two spawns.  `draws` yields (r10, k) given n_empty."""
    board = np.zeros((4, 4), dtype=np.int32)
    for _ in range(2):
        r10, k = draws(empty_count(board))
        spawn_injected(board, r10, k)
    return board


# ------------------------------------------------------------------ n-tuple encoders (a-6 .. a-8)

def f_2(x):
    """r_learning.py:17-20 — 12 vertical then 12 horizontal pairs, index 16*first + second."""
    vert = (x[0:3, :] * 16 + x[1:4, :]).ravel()
    horiz = (x[:, 0:3] * 16 + x[:, 1:4]).ravel()
    return np.concatenate([vert, horiz])


def f_3(x):
    """r_learning.py:24-31 — 8 vertical + 8 horizontal triples, then 4x9 L-shapes of each 2x2 window."""
    def enc(a, b, c):
        return (a * 256 + b * 16 + c).ravel()
    tl, tr = x[0:3, 0:3], x[0:3, 1:4]          # 2x2 window corners: top-left, top-right
    bl, br = x[1:4, 0:3], x[1:4, 1:4]          # bottom-left, bottom-right
    return np.concatenate([
        enc(x[0:2, :], x[1:3, :], x[2:4, :]),
        enc(x[:, 0:2], x[:, 1:3], x[:, 2:4]),
        enc(bl, br, tr),                        # missing top-left      (x_ex_00)
        enc(tl, bl, br),                        # missing top-right     (x_ex_01)
        enc(tl, tr, br),                        # missing bottom-left   (x_ex_10)
        enc(tl, bl, tr),                        # missing bottom-right  (x_ex_11)
    ])


def _quads(x):
    """The 17 four-cell features shared by f_4/f_5/f_6 (r_learning.py:40-44): columns, rows, 2x2 squares."""
    cols = x[0, :] * 4096 + x[1, :] * 256 + x[2, :] * 16 + x[3, :]
    rows = x[:, 0] * 4096 + x[:, 1] * 256 + x[:, 2] * 16 + x[:, 3]
    sq = (x[0:3, 0:3] * 4096 + x[1:4, 0:3] * 256 + x[0:3, 1:4] * 16 + x[1:4, 1:4]).ravel()
    return [cols.ravel(), rows.ravel(), sq]


def _crosses(x):
    """4 plus-shaped 5-cell features around the middle cells (r_learning.py:51-52):
    centre, up, left, down, right from most to least significant nibble."""
    c = x[1:3, 1:3]
    up, left, down, right = x[0:2, 1:3], x[1:3, 0:2], x[2:4, 1:3], x[1:3, 2:4]
    return (c * 65536 + up * 4096 + left * 256 + down * 16 + right).ravel()


def _hexes(x):
    """12 six-cell features, base 14 on min(tile, 13) (r_learning.py:63-68)."""
    y = np.minimum(x, 13)
    m = [14 ** 5, 14 ** 4, 14 ** 3, 14 ** 2, 14, 1]
    tall = (m[0] * y[0:2, 0:3] + m[1] * y[1:3, 0:3] + m[2] * y[2:4, 0:3]
            + m[3] * y[0:2, 1:4] + m[4] * y[1:3, 1:4] + m[5] * y[2:4, 1:4]).ravel()
    wide = (m[0] * y[0:3, 0:2] + m[1] * y[0:3, 1:3] + m[2] * y[0:3, 2:4]
            + m[3] * y[1:4, 0:2] + m[4] * y[1:4, 1:3] + m[5] * y[1:4, 2:4]).ravel()
    return [tall, wide]


def f_4(x):
    return np.concatenate(_quads(x))


def f_5(x):
    return np.concatenate(_quads(x) + [_crosses(x)])


def f_6(x):
    return np.concatenate(_quads(x) + [_crosses(x)] + _hexes(x))


FEATURES = {2: f_2, 3: f_3, 4: f_4, 5: f_5, 6: f_6}           # r_learning.py:87
NUM_FEAT = {2: 24, 3: 52, 4: 17, 5: 21, 6: 33}               # r_learning.py:88
# (group sizes, slots per feature) in weight_signature order, r_learning.py:136-149
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


def feature_offsets(n):
    """Flat-table offset of each feature (feature-major, weight_signature group order)."""
    sizes = feature_sizes(n)
    return np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64), int(np.sum(sizes))


def d4_images(board):
    """The 8 images visited by QAgent.update (r_learning.py:207-214), in its order:
    x, T(x), then x <- rot90(T(T(x))) = rot90(x), repeated four times."""
    images = []
    cur = board
    for _ in range(4):
        images.append(cur)
        t = np.transpose(cur)
        images.append(t)
        cur = np.rot90(np.transpose(t))
    return images


# ------------------------------------------------------------------ learner (a-9 .. a-13)

class Agent:
    """QAgent restated (r_learning.py:85-252) — weights are nested Python lists of float64."""

    def __init__(self, n=4, alpha=0.25, weights=None):
        self.n = n
        self.alpha = alpha
        self.num_feat = NUM_FEAT[n]
        self.features = FEATURES[n]
        sizes = feature_sizes(n)
        if weights is None:                        # init_weights, r_learning.py:136-149
            self.weights = [(np.random.random(s) / 100).tolist() for s in sizes]
        else:                                      # flat array in group order -> list of rows
            flat = np.asarray(weights)
            offs, total = feature_offsets(n)
            assert flat.shape == (total,)
            self.weights = [flat[o:o + s].astype(np.float64).tolist() for o, s in zip(offs, sizes)]

    def flat_weights(self):
        return np.concatenate([np.asarray(w, dtype=np.float64) for w in self.weights])

    def evaluate(self, board, score=None):
        """r_learning.py:202-203: left-to-right float64 sum of one weight per feature."""
        total = 0
        for i, f in enumerate(self.features(board)):
            total = total + self.weights[i][f]
        return total

    def update(self, board, dw):
        """r_learning.py:207-214: += dw at every feature slot of the 8 symmetric images."""
        for image in d4_images(board):
            for i, f in enumerate(self.features(image)):
                self.weights[i][f] += dw

    def best_move(self, board, score):
        """Greedy choice, r_learning.py:229-237: first maximum over the directions that change the board."""
        action, best_value, best_board, best_score = 0, -np.inf, None, None
        for d in range(4):
            nb, ns, moved = pre_move(board, score, d)
            if moved:
                v = self.evaluate(nb)
                if v > best_value:
                    action, best_value, best_board, best_score = d, v, nb, ns
        return action, best_value, best_board, best_score

    def episode(self, draws, trace=None, max_steps=None):
        """One self-play game with online TD(0), r_learning.py:224-252.

        `draws(n_empty) -> (r10, k)` supplies the spawn draws.  If `trace` is a list, one dict
        per move is appended.  Returns (board, score, n_moves)."""
        board = new_game(draws)
        score, moves = 0, 0
        state, old_label = None, 0
        while not game_over(board):
            if max_steps is not None and moves >= max_steps:
                return board, score, moves
            action, best_value, best_board, best_score = self.best_move(board, score)
            dw = None
            if state is not None:
                dw = (best_score - score + best_value - old_label) * self.alpha / self.num_feat
                self.update(state, dw)
            if trace is not None:
                trace.append(dict(board=board.copy(), action=action, reward=best_score - score,
                                  value=best_value, dw=dw))
            board, score = np.array(best_board), best_score
            moves += 1
            state, old_label = board.copy(), best_value
            r10, k = draws(empty_count(board))
            spawn_injected(board, r10, k)
        dw = -old_label * self.alpha / self.num_feat
        self.update(state, dw)
        if trace is not None:
            trace.append(dict(board=board.copy(), action=-1, reward=0, value=0.0, dw=dw))
        return board, score, moves


def random_episode(draws, picks):
    """Env-only game with a uniformly random *valid* direction (BASELINE config 2).
    picks(n_valid) -> index into the list of changing directions."""
    board = new_game(draws)
    score, moves = 0, 0
    while not game_over(board):
        cands = []
        for d in range(4):
            nb, ns, moved = pre_move(board, score, d)
            if moved:
                cands.append((nb, ns))
        nb, ns = cands[picks(len(cands))]
        board, score = np.array(nb), ns
        moves += 1
        r10, k = draws(empty_count(board))
        spawn_injected(board, r10, k)
    return board, score, moves
