"""`Game` — the reference's single-board object (game2048/game_logic.py:45-269), same attributes and methods, with the
board arithmetic done by the HIP kernels through the C ABI (include/g2048.h): moving (`pre_move`, `make_move`,
`Game.table`), the terminal test and the counts come from the device.  What stays on the host is the record of one
game (moves, tiles, history, pickling, printing) and the look-ahead driver, which are not compute.

There is no CPU fallback: the first use of a board operation raises G2048Error if lib2048_hip.so or a GPU is missing.
"""
import threading

from .start import *  # noqa: F401,F403  (the reference's modules star-import start.py the same way)
from .start import GAME_PANE, Thread, np, pickle, random

from . import _lib
from .engine import Engine


# basic evaluation methods (game_logic.py:5-10)
def random_eval(row, score):
    return np.random.random()


def score_eval(row, score):
    return score


class _Device:
    """One tiny device context (batch 1, no weight table) shared by all Game objects of the process; calls are
    serialised with a lock because the UI drivers run games from daemon threads (game_logic.py:199-200)."""

    def __init__(self):
        self.lock = threading.Lock()
        self.eng = None
        self.key = None
        self.moves = None

    def engine(self):
        backend = _lib.default_backend()                      # (the backend is an explicit, per-process choice: G2048_BACKEND)
        if self.eng is None or self.eng.backend != backend:
            self.eng, self.key = Engine(1, n=0, seed=2048, backend=backend), None
        return self.eng

    def move_all(self, row):
        """(after[4,4,4], reward[4], changed bits) of one board, cached for the 4 pre_move calls that follow."""
        b = np.ascontiguousarray(row, dtype=np.uint8)
        key = b.tobytes()
        with self.lock:
            if key != self.key:
                after, reward, changed = self.engine().boards_move_all(b.reshape(1, 4, 4))
                self.key, self.moves = key, (after[0].astype(np.int32), reward[0], int(changed[0]))
            return self.moves

    def terminal(self, row):
        b = np.ascontiguousarray(row, dtype=np.uint8)
        with self.lock:
            eng = self.engine()
            eng.set_boards(b.reshape(1, 4, 4))
            over, n_empty, n_pairs = eng.terminal()
            return bool(over[0]), int(n_empty[0]), int(n_pairs[0])


_DEVICE = _Device()


def create_table():
    """The 65 536-row move dictionary of the reference (game_logic.py:18-39), computed by k_move_all on the device:
    (a, b, c, d) -> (row after moving left, score, changed)."""
    keys = np.arange(65536)
    lines = np.stack([(keys >> 12) & 15, (keys >> 8) & 15, (keys >> 4) & 15, keys & 15], axis=1).astype(np.uint8)
    boards = np.zeros((65536, 4, 4), np.uint8)
    boards[:, 0, :] = lines
    eng = Engine(65536, n=0)
    eng.set_boards(boards)
    after, reward, changed = eng.move_all()
    eng.close()
    table = {}
    for k in range(65536):
        line = tuple(int(v) for v in lines[k])
        table[line] = (tuple(int(v) for v in after[k, 0, 0]), int(reward[k, 0]), bool(changed[k] & 1))
    print('table of moves created')
    return table


class _LazyTable:
    """Game.table is built on first access instead of at import time (the reference builds it at class creation,
    game_logic.py:51) so that importing the module does not need a GPU."""

    def __init__(self):
        self.value = None

    def __get__(self, obj, owner):
        if self.value is None:
            self.value = create_table()
        return self.value


class Game:
    """One 4x4 game.  `row` is an int32[4,4] array of log2 tiles (0 = empty), `score` the running score, `odometer`
    the number of moves made, `moves` / `tiles` the record that `replay` needs (game_logic.py:45-66)."""

    actions = {0: 'left', 1: 'up', 2: 'right', 3: 'down'}          # game_logic.py:50
    table = _LazyTable()                                            # game_logic.py:51
    counter = 0                                                     # calls to pre_move ("shuffles"), game_logic.py:52,137
    save_file = 'saved_game.pkl'

    def __init__(self, score=0, row=None, file=None):
        self.score, self.odometer = score, 0
        self.moves, self.tiles, self.history = [], [], {}
        self.file = file or Game.save_file
        if row is not None:
            self.row = np.array(row, dtype=np.int32)
            self.starting_position = row
            return
        self.row = np.zeros((4, 4), dtype=np.int32)                 # a fresh game: two tiles, not recorded in `tiles`
        self.new_tile()
        self.new_tile()
        self.tiles = []
        self.starting_position = self.row.copy()

    # ---- identity, persistence, display

    def copy(self):
        return Game(self.score, self.row)

    def __eq__(self, other):
        return np.array_equal(self.row, other.row)

    def save_game(self, file=None):                                 # game_logic.py:77-80
        with open(file or self.file, 'wb') as f:
            pickle.dump(self, f, -1)

    @staticmethod
    def load_game(file=save_file):                                  # game_logic.py:82-86
        with open(file, 'rb') as f:
            return pickle.load(f)

    def __str__(self):                                              # layout of game_logic.py:91-94
        lines = []
        for board_row in self.row:
            cells = []
            for v in board_row:
                face = (1 << int(v)) if v else 0
                cells.append(str(face) + '\t' * (3 if face >= 1000 else 4))
            lines.append(''.join(cells))
        lines.append(f' score = {self.score} moves = {self.odometer} reached {1 << int(np.max(self.row))}')
        return '\n'.join(lines)

    # ---- board queries (device: k_terminal)

    @staticmethod
    def empty(row):
        """Empty cells in row-major order (game_logic.py:96-99) — a host list, it only feeds random.choice."""
        rr, cc = np.nonzero(np.asarray(row) == 0)
        return list(zip(rr, cc))

    @staticmethod
    def empty_count(row):                                           # game_logic.py:101-103
        return _DEVICE.terminal(row)[1]

    @staticmethod
    def adjacent_pair_count(row):                                   # game_logic.py:105-107
        return _DEVICE.terminal(row)[2]

    def game_over(self, row):                                       # game_logic.py:109-110
        return _DEVICE.terminal(row)[0]

    # ---- the random tile (game_logic.py:112-121): draw order is tile first, then cell

    def create_new_tile(self, row):
        cells = self.empty(row)
        tile = 2 if random.randrange(10) == 0 else 1
        return tile, random.choice(cells)

    def new_tile(self):
        tile, position = self.create_new_tile(self.row)
        self.row[position] = tile
        self.tiles.append((tile, position))

    # ---- moves (device: k_move_all)

    @staticmethod
    def _left(row, score):                                          # game_logic.py:123-134 = direction 0
        after, reward, changed = _DEVICE.move_all(row)
        return after[0].copy(), score + int(reward[0]), bool(changed & 1)

    def pre_move(self, row, score, direction):                      # game_logic.py:136-142
        Game.counter += 1
        after, reward, changed = _DEVICE.move_all(row)
        return after[direction].copy(), score + int(reward[direction]), bool((changed >> direction) & 1)

    def make_move(self, direction):                                 # game_logic.py:144-148
        self.row, self.score, changed = self.pre_move(self.row, self.score, direction)
        self.odometer += 1
        self.moves.append(direction)
        return changed

    # ---- greedy / look-ahead play on top of an estimator callable (row, score) -> value

    def _find_best_move(self, estimator, depth, width, since_empty):
        """First maximum over the directions that change the board (game_logic.py:150-161)."""
        best = (0, None, None)
        best_value = -np.inf
        engine = getattr(getattr(estimator, '__self__', None), 'engine', None)
        if depth > 0 and engine is not None and getattr(estimator, '__name__', '') == 'evaluate':
            # the estimator is a device agent: the look-ahead trees of the four candidates are expanded, evaluated and reduced
            # on the device (g2048_boards_look_forward, csrc/lookahead.hip) instead of node by node; their chance nodes are
            # keyed by a fresh salt from `random` (the reference draws them from `random` too)
            after, reward, changed = _DEVICE.move_all(self.row)
            dirs = [d for d in range(4) if (changed >> d) & 1]
            if not dirs:
                return best
            salt = np.array([[random.getrandbits(64), random.getrandbits(64)]] * len(dirs), np.uint64)
            values = engine.boards_look_forward(after[dirs].astype(np.uint8), depth, width, since_empty, salt)
            Game.counter += 4
            for d, value in zip(dirs, values):
                if value > best_value:
                    best_value, best = value, (d, after[d].copy(), self.score + int(reward[d]))
            return best
        for direction in range(4):
            cand_row, cand_score, changed = self.pre_move(self.row, self.score, direction)
            if not changed:
                continue
            value = self.look_forward(estimator, cand_row, cand_score, depth=depth, width=width, since_empty=since_empty)
            if value > best_value:
                best_value, best = value, (direction, cand_row, cand_score)
        return best

    def _move_on(self, best_dir, best_row, best_score):             # game_logic.py:163-167
        self.moves.append(best_dir)
        self.odometer += 1
        self.row, self.score = best_row, best_score
        self.new_tile()

    def _steps(self, estimator, limit_tile, depth, width, since_empty):
        """Common loop of trial_run / generate_run: yields the chosen direction before it is played."""
        while not self.game_over(self.row):
            if limit_tile and np.max(self.row) >= limit_tile:
                return
            choice = self._find_best_move(estimator, depth, width, since_empty)
            yield choice[0]
            self._move_on(*choice)

    def trial_run(self, estimator, limit_tile=0, step_limit=100000, depth=0, width=1, since_empty=0, verbose=False):
        """Play to the end, or until a tile >= limit_tile is on the board, or step_limit moves (game_logic.py:170-183)."""
        if verbose:
            print('Starting position:')
            print(self)
        while self.odometer < step_limit and not self.game_over(self.row):
            if limit_tile and np.max(self.row) >= limit_tile:
                break
            choice = self._find_best_move(estimator, depth, width, since_empty)
            self._move_on(*choice)
            if verbose:
                print(f'On {self.odometer} we moved {Game.actions[choice[0]]}')
                print(self)

    def generate_run(self, estimator, limit_tile=0, depth=0, width=1, since_empty=16):
        """Generator for show.py's watch mode (game_logic.py:203-211): yields (game, direction) before each move."""
        for direction in self._steps(estimator, limit_tile, depth, width, since_empty):
            yield self, direction

    def trial_run_for_thread(self, estimator, depth=0, width=1, since_empty=0, stopper=None):
        """Dash "Agent Play" worker (game_logic.py:186-197): records `history` and stops when the pane is re-used."""
        parent, this_thread = stopper['parent'], stopper['n']
        while GAME_PANE[parent]['id'] == this_thread:
            if self.game_over(self.row):
                self.history[self.odometer] = (self.row.copy(), self.score, -1)
                self.moves.append(-1)
                return
            choice = self._find_best_move(estimator, depth, width, since_empty)
            self.history[self.odometer] = (self.row.copy(), self.score, choice[0])
            self._move_on(*choice)

    def thread_trial(self, *args, **kwargs):                        # game_logic.py:199-200
        Thread(target=self.trial_run_for_thread, args=args, kwargs=kwargs, daemon=True).start()

    def look_forward(self, estimator, row, score, depth, width, since_empty):
        """Sampled expectimax (game_logic.py:214-243).  Depth 0 — what the learner and show.py use — is one estimator
        call.  Deeper: if the board has fewer than `since_empty` empty cells, average over `width` sampled new tiles
        of max(0, best child value), a dead position counting -100."""
        if depth == 0:
            return estimator(row, score)
        n_empty = self.empty_count(row)
        if n_empty >= since_empty:
            return estimator(row, score)
        samples = random.sample(self.empty(row), min(width, n_empty))
        total = 0
        for cell in samples:
            child = row.copy()
            child[cell] = 2 if random.randrange(10) == 0 else 1
            if self.game_over(child):
                outcome = -100
            else:
                outcome = -np.inf
                for direction in range(4):
                    nxt_row, nxt_score, changed = self.pre_move(child, score, direction)
                    if changed:
                        outcome = max(outcome, self.look_forward(estimator, nxt_row, nxt_score, depth=depth - 1,
                                                                 width=width, since_empty=since_empty))
            total += max(outcome, 0)
        return total / len(samples)

    def replay(self, verbose=True):
        """Re-run the recorded game; returns {move index: (board, score, direction)} (game_logic.py:246-269)."""
        shadow = Game(row=self.starting_position)
        if verbose:
            print('Starting position:')
            print(shadow)
        chain = {}
        for i in range(self.odometer):
            direction = self.moves[i]
            tile, cell = self.tiles[i]
            chain[i] = (shadow.row.copy(), shadow.score, direction)
            if verbose:
                print(i, tile, cell)
            shadow.make_move(direction)
            shadow.row[cell] = tile
            if verbose:
                print(f'On {shadow.odometer} we move = {Game.actions[direction]}, new tile = {tile} at position = {cell}')
                print(shadow)
        if verbose:
            print('no more moves possible, final position')
        chain[self.odometer] = (self.row.copy(), self.score, self.moves[self.odometer])
        chain[self.odometer + 1] = (None, None, -1)
        return chain


__all__ = [n for n in dir() if not n.startswith('_')]
