"""`QAgent` (alias `Q_agent`) — the reference's n-tuple TD(0) learner (game2048/r_learning.py:85-406) on the device.

Same constructor, attributes and methods as the reference; the table lives in HBM as one flat fp32 array
(feature-major, weight_signature group order) and every piece of the hot path — f_n encoding, evaluate, greedy
choice, update over the 8 symmetries, whole episodes — runs in the HIP kernels behind include/g2048.h.
Added keyword arguments: `batch` (concurrent episodes used by train_run / trial; 1 = the reference's one game at a
time), `seed`, `device`.

Batched TD (batch > 1) is synchronous: in one board-step every lane chooses with the same table, then all
(state, dw) records are applied.  `rule` says how:
  'mean' (default for batch > 1): a slot moves by the MEAN of the dw that target it in the step, with the reference's
         alpha unchanged — stable at any batch size and the rule that learns (n = 4, 65 536 lanes: mean score 50 000,
         2048 in 93 % of games after 45 s of training);
  'sum'  the reference's arithmetic (every dw is added): one slot can receive the adds of all 8 images of every lane
         in the same step, so alpha is divided accordingly: alpha_device = alpha * num_feat / (8 * batch).
Batch 1 always uses 'sum' with the reference's alpha and reproduces the reference's trajectories.
"""
from .game import *  # noqa: F401,F403  (r_learning.py:3 star-imports game_logic the same way)
from .game import Game, np, pickle, time, deque, load_s3, save_s3, Logger, AGENT_PANE, RUNNING, dash_intervals

from . import _lib
from .engine import Engine, NUM_FEAT, feature_layout


def check_thread(parent, benchmark):               # r_learning.py:6-13: "is the browser still there" heartbeat
    now = time.time()
    if (now - benchmark) > 2 * dash_intervals['check_run']:
        if RUNNING[parent] == 0:
            return 0
        RUNNING[parent] = 0
        return now
    return benchmark


# ---- n-tuple encoders f_2 .. f_6 (r_learning.py:17-69), computed by k_features on the device

_FEATURE_ENGINES = {}


def _features(n, x):
    key = (_lib.default_backend(), n)
    eng = _FEATURE_ENGINES.get(key)
    if eng is None:
        eng = _FEATURE_ENGINES[key] = _TableFree(n)
    return eng(x)


class _TableFree:
    """A 1-lane context used only for g2048_features (the table stays untouched: the n=6 one is 383 MB, so the
    context is created once per n and kept)."""

    def __init__(self, n):
        self.eng = Engine(1, n=n)

    def __call__(self, x):
        self.eng.set_boards(np.asarray(x, dtype=np.uint8).reshape(1, 4, 4))
        return self.eng.features()[0].astype(np.int64)


def f_2(x):
    return _features(2, x)


def f_3(x):
    return _features(3, x)


def f_4(x):
    return _features(4, x)


def f_5(x):
    return _features(5, x)


def f_6(x):
    return _features(6, x)


class _Stopper:
    """The cooperative stop protocol of the Dash workers (r_learning.py:284-290): the pane's current worker id and the
    browser heartbeat."""

    def __init__(self, stopper, key='a'):
        self.parent, self.me = (stopper['parent'], stopper[key]) if stopper else (None, None)
        self.mark = time.time()

    def poll(self):
        if self.parent is None:
            return 'go'
        if AGENT_PANE[self.parent]['id'] != self.me:
            return 'stop'
        self.mark = check_thread(self.parent, self.mark)
        return 'go' if self.mark else 'abandoned'


class _Window:
    """Statistics of the games since the last 1000-episode report (r_learning.py:279-282, 297-309)."""

    def __init__(self):
        self.scores, self.last100 = [], deque(maxlen=100)
        self.reached = [0] * 7                  # games whose largest tile was 2^10 .. 2^16
        self.best, self.since = None, time.time()

    def add(self, game, top_tile):
        """Returns True if `game` is the best of the window so far."""
        self.scores.append(game.score)
        self.last100.append(game.score)
        if top_tile >= 10:
            self.reached[min(top_tile, 16) - 10] += 1
        if self.best is None or game.score > self.best.score:
            self.best = game
            return True
        return False

    def ma100(self):
        return int(np.mean(self.last100))

    def mean(self):
        return np.mean(self.scores)


GROUPS = {2: (24,), 3: (52,), 4: (17,), 5: (17, 4), 6: (17, 4, 12)}       # weight_signature, r_learning.py:136-149


class QAgent:

    feature_functions = {2: f_2, 3: f_3, 4: f_4, 5: f_5, 6: f_6}                                       # r_learning.py:87
    parameter_shape = {2: (24, 16 ** 2), 3: (52, 16 ** 3), 4: (17, 16 ** 4), 5: (21, 16 ** 5), 6: (33, 0)}     # :88

    def __init__(self, name='agent', config_file=None, storage='s3', console='web', log_file=None, n=4, alpha=0.25,
                 decay=0.75, decay_step=10000, low_alpha_limit=0.01, with_weights=True, batch=1, seed=2048, device=0,
                 rule=None, dist=None, epoch=64, comm='native'):
        """The reference's arguments (r_learning.py:90-91), then this build's: `batch` concurrent episodes on GPU `device`,
        `rule` ('sum' = the reference's update, 'mean' = per-slot mean, the default for batch > 1), and for multi-GPU
        training `dist` — an initialised torch.distributed module (one process per GPU: this agent is rank
        dist.get_rank(), its `batch` lanes are the rank's shard of world * batch episodes, the table is replicated and the
        accumulated weight deltas are all-reduced every `epoch` board-steps; comm = 'native' uses the C ABI's own RCCL
        all-reduce, 'torch' the process group's)."""
        # identity and I/O (r_learning.py:93-99)
        self.name, self.file, self.game_file = name, name + '.pkl', 'best_of_' + name + '.pkl'
        self.s3, self.log_file = storage == 's3', log_file
        to_console = console == 'local' or log_file is None
        self.print = print if to_console else Logger(log_file=log_file).add
        # hyper-parameters: a stored JSON config wins over the arguments (r_learning.py:101-110)
        stored = (load_s3(config_file) or {}) if config_file else {}
        given = dict(n=n, alpha=alpha, decay=decay, decay_step=decay_step, low_alpha_limit=low_alpha_limit)
        for key, value in given.items():
            setattr(self, key, stored.get(key, value))
        self.num_feat, self.size_feat = QAgent.parameter_shape[self.n]      # r_learning.py:112-114
        self.features = QAgent.feature_functions[self.n]
        # training state (r_learning.py:116-122)
        self.step, self.top_score, self.top_tile = 0, 0, 10
        self.top_game, self.train_history = None, []
        self.next_decay = self.decay_step
        # device side
        self.batch, self.seed, self.device = int(batch), int(seed), int(device)
        self._dist, self.epoch, self.comm = dist, int(epoch), comm
        self.rank = dist.get_rank() if dist is not None else 0
        self.world = dist.get_world_size() if dist is not None else 1
        self._sync = None
        self.rule = rule or ('mean' if self.batch > 1 else 'sum')
        self._engine = self._solo = None
        self._pending_weights = None
        self.weight_signature = None
        if with_weights:
            self.init_weights()

    def __str__(self):
        return f'Agent {self.name}, n={self.n}\ntrained for {self.step} episodes, top score = {self.top_score}'

    # ---- device contexts

    @property
    def engine(self):
        """The training batch (owns the weight table)."""
        if self._engine is None:
            self._engine = Engine(self.batch, n=self.n, seed=self.seed, lane0=self.rank * self.batch, device=self.device)
            if self.batch > 1 and self.rule == 'mean':
                self._engine.set_update_rule(1)
            if self._pending_weights is not None:
                self._engine.set_weights(self._pending_weights)
                self._pending_weights = None
        return self._engine

    @property
    def solo(self):
        """One lane over the same table: evaluate / update of single boards and QAgent.episode."""
        if self._solo is None:
            self._solo = Engine(1, seed=self.seed + 0x5010, lane0=1 << 40, share_table_of=self.engine)
            self._solo.set_auto_reset(False)
        return self._solo

    # ---- weights: a flat fp32 table on the device; `weights` shows it in the reference's list-of-rows form

    def init_weights(self):
        """U[0, 0.01) per slot (r_learning.py:136-149), drawn on the device."""
        seed = int(np.random.randint(0, 2 ** 31))
        if self.world > 1:                                    # every replica starts from rank 0's table (counter-based init)
            box = [seed]
            self._dist.broadcast_object_list(box, src=0)
            seed = box[0]
        self.engine.init_weights(seed=seed, scale=0.01)
        self.weight_signature = GROUPS[self.n]

    @property
    def weights(self):
        if self._engine is None and self._pending_weights is None:
            return None
        # (weights handed over before the first use of the device can be read back — and pickled — without one)
        flat = self._pending_weights if self._engine is None else self.engine.get_weights()
        offs, sizes = feature_layout(self.n, getattr(self._engine, 'backend', None))
        return [flat[o:o + s] for o, s in zip(offs, sizes)]

    @weights.setter
    def weights(self, rows):
        if rows is None:
            return
        flat = np.concatenate([np.asarray(r, dtype=np.float32).reshape(-1) for r in rows])
        if self._engine is None:
            self._pending_weights = flat
        else:
            self._engine.set_weights(flat)
        self.weight_signature = GROUPS[self.n]

    def list_to_np(self):
        """Weights as one float32 array per weight_signature group — the on-disk form (r_learning.py:151-158)."""
        rows = self.weights
        out, start = [], 0
        for d in self.weight_signature:
            out.append(np.stack(rows[start:start + d]).astype(np.float32))
            start += d
        return out

    def np_to_list(self):
        """Inverse of list_to_np (r_learning.py:160-164); a no-op here, rows are always views of the flat table."""

    # ---- persistence (r_learning.py:166-200): parameters and weights are stored separately in 's3' mode

    def __getstate__(self):
        state = {k: v for k, v in self.__dict__.items() if k not in ('_engine', '_solo', '_pending_weights', 'print', '_dist', '_sync', '_shared_best')}
        state['weights'] = self.list_to_np() if (self._engine is not None or self._pending_weights is not None) else None
        return state

    def __setstate__(self, state):
        groups = state.pop('weights', None)
        self.__dict__.update(state)
        self.__dict__.setdefault('batch', 1)
        self.__dict__.setdefault('seed', 2048)
        self.__dict__.setdefault('device', 0)
        self.__dict__.setdefault('rule', 'mean' if self.batch > 1 else 'sum')
        self.__dict__.update(rank=0, world=1)                 # a loaded agent is single-process until attach_dist
        self.__dict__.setdefault('epoch', 64)
        self.__dict__.setdefault('comm', 'native')
        self._engine = self._solo = self._pending_weights = self._dist = self._sync = None
        self.print = print
        self.features = QAgent.feature_functions[self.n]
        if groups is not None:
            self.weights = [row for g in groups for row in np.asarray(g)]

    def save_agent(self):
        if self.s3:
            nps = self.list_to_np()
            params = QAgent(name=self.name, with_weights=False)
            for key, value in self.__dict__.items():
                if key not in ('_engine', '_solo', '_pending_weights', '_dist', '_sync', '_shared_best'):
                    setattr(params, key, value)
            save_s3(params, 'a/' + self.file)
            save_s3(nps, 'weights/' + self.file)
        else:
            with open(self.file, 'wb') as f:
                pickle.dump(self, f, -1)

    def save_game(self, game):
        if self.s3:
            save_s3(game, 'g/' + self.game_file)
        else:
            game.save_game(self.game_file)

    @staticmethod
    def load_agent_local(file):
        with open(file, 'rb') as f:           # (the reference opens in text mode, r_learning.py:190 — a bug)
            return pickle.load(f)

    @staticmethod
    def load_agent(file):
        agent = load_s3(file)
        groups = load_s3(f'weights/{file[2:]}')
        agent.weights = [row for g in groups for row in np.asarray(g)]
        return agent

    # ---- value and update of one board (r_learning.py:202-214)

    def evaluate(self, row, score=None):
        return float(self.engine.boards_evaluate(np.asarray(row, dtype=np.uint8).reshape(1, 4, 4))[0])

    def update(self, row, dw):
        self.engine.update(np.asarray(row, dtype=np.uint8).reshape(1, 4, 4), np.array([dw], np.float32))

    # ---- one self-play game with online TD(0) (r_learning.py:224-252), on the solo lane

    def episode(self):
        solo = self.solo
        solo.reset()
        solo.clear_carry()
        game = Game(row=solo.get_boards()[0])
        game.starting_position = game.row.copy()
        while True:
            solo.td_steps(self.alpha, 1)
            lm = int(solo.last_move()[0])
            if not lm & 4:
                break
            game.moves.append(lm & 3)
            game.odometer += 1
            if lm & (1 << 10):
                cell = (lm >> 4) & 15
                game.tiles.append(((lm >> 8) & 3, (cell >> 2, cell & 3)))
            if lm & (1 << 11):
                break
        game.row = solo.get_boards()[0].astype(np.int32)
        game.score = int(solo.get_scores()[0])
        game.moves.append(-1)
        self.step += 1
        return game

    def _display_lr(self):
        self.print(f'episode = {self.step + 1}, current learning rate = {round(self.alpha, 4)}:')

    def decay_alpha(self):                                           # r_learning.py:257-262
        self.alpha = round(max(self.alpha * self.decay, self.low_alpha_limit), 4)
        self.next_decay = self.step + self.decay_step
        self.print('------')
        self._display_lr()
        self.print('------')

    # ---- training loop (r_learning.py:269-346)

    def train_run(self, num_eps=100000, add_weights='already', saving=True, stopper=None):
        if add_weights == 'add':
            self.init_weights()
        elif add_weights != 'already':
            self.print('loading weights ...')
            groups = load_s3(add_weights)
            self.weights = [row for g in groups for row in np.asarray(g)]
        if self.batch > 1:
            return self._train_run_batched(num_eps, saving, stopper)
        watch = _Stopper(stopper)
        window = _Window()
        began = time.time()
        self.print(f'Agent {self.name} training session started, current step = {self.step}')
        self.print('Agent will be saved every 1000 episodes and on STOP command')
        first = self.step + 1
        for i in range(first, first + num_eps + 1):                  # (the reference also plays num_eps + 1 games, :284)
            verdict = watch.poll()
            if verdict == 'stop':
                break
            if verdict == 'abandoned':
                return
            if self.step > self.next_decay and self.alpha > self.low_alpha_limit:
                self.decay_alpha()
            game = self.episode()
            top = int(np.max(game.row))
            if window.add(game, top) and game.score > self.top_score:       # best of this window and of the agent
                self.top_game, self.top_score = game, game.score
                self.print(f'\nnew best game at episode {i}!\n{game}\n')
                if saving:
                    self.save_game(game)
                    self.print(f'game saved at {self.game_file}')
            if top > self.top_tile:                                  # a new largest tile also decays alpha (r_learning.py:311-313)
                self.top_tile = top
                self.decay_alpha()
            if i % 100 == 0:
                self.train_history.append(window.ma100())
                self.print(f'episode {i}: score {game.score} reached {1 << top} ma_100 = {self.train_history[-1]}')
            if i % 1000 == 0:
                self._report_1000(i, window.mean(), window.reached, window.best, time.time() - window.since, saving)
                window = _Window()
        self._finish(began, saving)

    def _report_1000(self, i, average, reached, best, seconds, saving):     # r_learning.py:318-341
        self.print('\n------')
        self.print(f'{round(seconds / 60, 2)} min')
        self.print(f'episode = {i}')
        self.print(f'average over last 1000 episodes = {average}')
        total = max(1, sum(reached)) if sum(reached) > 1000 else 1000
        for j in range(7):
            r = sum(reached[j:]) / (total / 100)
            if r:
                self.print(f'{1 << (j + 10)} reached in {r} %')
        if best is not None:
            self.print('best of last 1000:')
            self.print(str(best))
        if self.top_game is not None:
            self.print('best of this Agent:')
            self.print(str(self.top_game))
        self._display_lr()
        self.print('------\n')
        if saving:
            self.save_agent()
            self.print(f'agent saved in {self.file}')

    def _finish(self, global_start, saving):                         # r_learning.py:342-346
        total_time = int(time.time() - global_start)
        self.print(f'Total time = {total_time // 60} min {total_time % 60} sec')
        if saving:
            self.save_agent()
            self.print(f'{self.name} saved at step {self.step} in {self.file}\n------------------------\n')

    def attach_dist(self, dist, device=None):
        """Make a loaded (or single-process) agent one rank of a multi-GPU job; call before the engine exists."""
        assert self._engine is None, 'attach_dist before the first use of the device'
        self._dist, self.rank, self.world = dist, dist.get_rank(), dist.get_world_size()
        if device is not None:
            self.device = int(device)

    def _epoch_sync(self):
        """The per-epoch exchange of the accumulated weight deltas (2048_amd/parallel.py), built on first use."""
        if self.world == 1:
            return None
        if self._sync is None:
            from . import parallel
            sync, _ = parallel.make_sync(self.engine, self._dist, self.rank, self.world, rule=self.rule, comm=self.comm,
                                         log=self.print)
            sync.begin()
            self._sync = sync
        return self._sync

    def device_alpha(self, lanes=None):
        """The batch rule: alpha for g2048_td_steps when `lanes` episodes learn concurrently."""
        lanes = self.batch * self.world if lanes is None else lanes
        if lanes == 1 or self.rule == 'mean':
            return self.alpha
        return self.alpha * self.num_feat / (8.0 * lanes)

    WATCHED_LANES, LOG_CAPACITY = 1024, 16384

    def _game_from_log(self, eng, lane, slot, length, score, verify=False):
        """A reference `Game` (moves, tiles, starting_position, final row, score, odometer) from a device game record.
        verify: also replay the record move by move (Game.make_move) and check that it leads to the recorded end."""
        words, start = eng.log_game(lane, slot)
        game = Game(score=score, row=eng.log_final(lane, slot).astype(np.int32))
        game.starting_position = start.astype(np.int32)
        game.odometer = length
        for lm in words[:length]:
            lm = int(lm)
            game.moves.append(lm & 3)
            if lm & (1 << 10):
                game.tiles.append(((lm >> 8) & 3, (((lm >> 4) & 15) >> 2, ((lm >> 4) & 15) & 3)))
        game.moves.append(-1)
        if verify:
            check = Game(row=start.astype(np.int32))
            for lm in words[:length]:
                lm = int(lm)
                check.make_move(lm & 3)
                if lm & (1 << 10):
                    check.row[((lm >> 4) & 15) >> 2, ((lm >> 4) & 15) & 3] = (lm >> 8) & 3
            assert check.score == score and np.array_equal(check.row, game.row), 'device game record does not replay to its end'
        return game

    def _collect_best_games(self, eng, seen, saving):
        """Look at the games the watched lanes finished since the last call; keep the best one as top_game."""
        meta = eng.log_meta()
        best = None
        for lane in np.nonzero(meta[:, 2] != seen)[0]:
            slot = int(meta[lane, 0]) ^ 1                      # the slot that just finished
            length, score, flags = int(meta[lane, 3 + 2 * slot]), int(meta[lane, 4 + 2 * slot]), int(meta[lane, 7])
            if length and not (flags >> slot) & 5 and (best is None or score > best[0]):
                best = (score, int(lane), slot, length)
        seen[:] = meta[:, 2]
        if best is not None and (self.top_game is None or best[0] > self.top_game.score):
            self.top_game = self._game_from_log(eng, best[1], best[2], best[3], best[0])
            self.print(f'\nnew best recorded game at episode {self.step}!\n{self.top_game}\n')
            if saving:
                self.save_game(self.top_game)
                self.print(f'game saved at {self.game_file}')

    def _train_run_batched(self, num_eps, saving, stopper, chunk=64):
        """train_run on `batch` concurrent episodes: the same schedule and logs, driven by the device's episode
        counters; averages are over the games that finished in each reporting window.  `top_score` is the best score
        of all lanes; `top_game` (a replayable Game, as the reference keeps) is the best game among the first
        WATCHED_LANES lanes, whose moves and tiles the device records."""
        eng = self.engine
        eng.set_auto_reset(True)
        sync = self._epoch_sync()
        chief = self.rank == 0
        own_print = self.print
        if not chief:
            saving = False                                    # rank 0 prints, keeps the best game and saves
            self.print = lambda *a, **k: None
        if sync is not None:
            from . import parallel
            chunk = self.epoch

            def job_stats():                                  # the counters of all ranks: every rank takes the same decisions
                return parallel.reduce_stats(eng.stats(), self._dist, device=f'cuda:{self.device}' if self._dist.get_backend() == 'nccl' else 'cpu')
        else:
            job_stats = eng.stats
        watched = min(self.batch, self.WATCHED_LANES)
        eng.log_enable(watched, self.LOG_CAPACITY)
        seen = np.zeros(watched, np.uint32)
        if stopper:
            parent, this_thread = stopper['parent'], stopper['a']
        global_start = start = benchmark_time = time.time()
        self.print(f'Agent {self.name} training session started, current step = {self.step}, {self.batch} concurrent episodes'
                   + (f' on each of {self.world} GPUs, weight deltas exchanged every {chunk} steps' if sync is not None else ''))
        eng.stats_reset()
        base = self.step
        last = job_stats()
        mark100, mark1000 = dict(last), dict(last)
        next100 = (self.step // 100 + 1) * 100
        next1000 = (self.step // 1000 + 1) * 1000
        target = self.step + num_eps + 1
        while self.step < target:
            if stopper:
                if AGENT_PANE[parent]['id'] != this_thread:
                    break
                benchmark_time = check_thread(parent, benchmark_time)
                if not benchmark_time:
                    return
            # r_learning.py:293-294 looks at the episode counter before every game; here it jumps by thousands per chunk,
            # so every decay_step boundary that was passed decays once, as the per-episode check would have
            while self.step > self.next_decay and self.alpha > self.low_alpha_limit:
                due = self.next_decay
                self.decay_alpha()
                # the reference's test `self.step > self.next_decay` (r_learning.py:293) first holds at step = due + 1, and
                # decay_alpha then sets next_decay = step + decay_step (:259): the schedule slips by one episode per decay
                self.next_decay = due + 1 + self.decay_step
            eng.td_steps(self.device_alpha(), chunk)
            if sync is not None:
                sync.all_reduce()
            st = job_stats()
            self.step = base + st['episodes']
            self.top_score = max(self.top_score, st['best_score'])
            self._collect_best_games(eng, seen, saving and sync is None)
            if sync is not None:
                self._share_best_game(saving)
            top = max([t for t, cnt in enumerate(st['max_tile']) if cnt] or [0])
            if top > self.top_tile:
                # r_learning.py:311-313 decays at every game that sets a new maximum tile; a chunk holds thousands of games, so
                # every tile level above the old maximum that some game of the chunk ended on counts as one such game
                levels = [t for t in range(self.top_tile + 1, top + 1) if st['max_tile'][t] - last['max_tile'][t] > 0] or [top]
                self.top_tile = top
                for _ in levels:
                    self.decay_alpha()
            last = st
            while self.step >= next100:
                done = st['episodes'] - mark100['episodes']
                ma = int((st['score_sum'] - mark100['score_sum']) / max(1, done))
                self.train_history.append(ma)
                self.print(f'episode {next100}: ma_100 = {ma} (mean of the {done} games finished since the last report)')
                mark100, next100 = dict(st), next100 + 100 * max(1, (self.step - next100) // 100 + 1)
            if self.step >= next1000:
                done = st['episodes'] - mark1000['episodes']
                hist = [a - b for a, b in zip(st['max_tile'], mark1000['max_tile'])]
                reached = [sum(hist[10 + j:10 + j + 1]) for j in range(6)] + [sum(hist[16:])]
                average = (st['score_sum'] - mark1000['score_sum']) / max(1, done)
                self._report_1000(self.step, average, [r * 1000 / max(1, done) for r in reached], None, time.time() - start, saving)
                start, mark1000, next1000 = time.time(), dict(st), (self.step // 1000 + 1) * 1000
        self._finish(global_start, saving)
        self.print = own_print                                # (a non-chief rank was silenced for the run only)

    def _share_best_game(self, saving):
        """Multi-rank run: every rank watches its own lanes; the best recorded game of the JOB becomes every rank's top_game
        (the rank that holds it ships the Game), so that rank 0 saves a game that matches the job-wide bookkeeping."""
        mine = self.top_game.score if self.top_game is not None else -1
        scores = [None] * self.world
        self._dist.all_gather_object(scores, int(mine))
        owner = int(np.argmax(scores))
        if scores[owner] < 0 or scores[owner] == getattr(self, '_shared_best', -1):
            return
        box = [self.top_game if self.rank == owner else None]
        self._dist.broadcast_object_list(box, src=owner)
        self._shared_best = scores[owner]
        if self.rank != owner:
            self.top_game = box[0]
        if self.rank == 0:
            if owner != 0:                                    # (rank 0's own games were announced by _collect_best_games)
                self.print(f'\nnew best recorded game at episode {self.step} (played on rank {owner})!\n{self.top_game}\n')
            if saving:
                self.save_game(self.top_game)
                self.print(f'game saved at {self.game_file}')

    # ---- evaluation harness (r_learning.py:348-406)

    @staticmethod
    def trial(estimator=None, agent_file=None, limit_tile=0, num=20, game_init=None, depth=0, width=1, since_empty=6,
              storage='s3', console='local', log_file=None, game_file=None, verbose=False, stopper=None):
        """`num` games played with `estimator` (or a stored agent), then the summary the reference prints
        (r_learning.py:348-406).  A device agent at depth 0 plays all games at once on the GPU."""
        display = print if console == 'local' else Logger(log_file=log_file).add
        if agent_file:
            display(f'Loading Agent from {agent_file} ...')
            agent = QAgent.load_agent(agent_file)
            estimator = agent.evaluate
            display(f'Trial run for {num} games, Agent = {agent.name}\n'
                    f'Looking forward: depth={depth}, width={width}, since_empty={since_empty}')
        began = time.time()
        owner = getattr(estimator, '__self__', None)
        on_device = (isinstance(owner, QAgent) and getattr(estimator, '__name__', '') == 'evaluate' and not verbose and not stopper)
        if on_device:
            # all `num` games at once, entirely on the device with its game records: depth 0 greedy, depth > 0 with every game's
            # look-ahead tree expanded and reduced in HBM (csrc/lookahead.hip)
            results = owner._trial_batched(num, game_init, depth=depth, width=width, since_empty=since_empty, limit_tile=limit_tile)
            for i, game in enumerate(results):
                display(f'game {i}, result {game.score}, moves {game.odometer}, achieved {1 << np.max(game.row)}')
        else:
            watch, results = _Stopper(stopper), []
            for i in range(num):
                verdict = watch.poll()
                if verdict == 'stop':
                    break
                if verdict == 'abandoned':
                    return
                t0 = time.time()
                game = game_init.copy() if game_init is not None else Game()
                game.trial_run(estimator, limit_tile=limit_tile, depth=depth, width=width, since_empty=since_empty,
                               verbose=verbose)
                display(f'game {i}, result {game.score}, moves {game.odometer}, achieved {1 << np.max(game.row)}, '
                        f'time = {(time.time() - t0):.2f}')
                results.append(game)
        if not results:
            return
        results.sort(key=lambda g: g.score, reverse=True)
        display(QAgent._trial_summary(results, time.time() - began, 0 if on_device else Game.counter))
        if game_file:
            if storage == 's3':
                save_s3(results[0], game_file)
            else:
                results[0].save_game(file=game_file)
            display(f'Best game saved at {game_file}\n------------------------\n')
        return results

    @staticmethod
    def _trial_summary(results, elapsed, shuffles):
        """The closing message of QAgent.trial (r_learning.py:378-399): best games, mean score, tile shares, timings."""
        top_tiles = np.array([1 << int(np.max(g.row)) for g in results])
        moves = max(1, sum(g.odometer for g in results))
        lines = ['', 'Best games:'] + [f'{g}\n' for g in results[:3]]
        lines.append(f'average score of {len(results)} runs = {np.average([g.score for g in results])}')
        lines += [f'{limit} reached in {(top_tiles >= limit).mean() * 100}%' for limit in (16384, 8192, 4096, 2048, 1024)]
        lines.append(f'total time = {round(elapsed, 2)}')
        lines.append(f'average time per move = {round(elapsed / moves * 1000, 2)} ms')
        if shuffles:
            lines.append(f'total number of shuffles = {shuffles}')
            lines.append(f'time per shuffle = {round(elapsed / shuffles * 1000, 2)} ms')
        return '\n'.join(lines)

    TRIAL_LOG_BYTES = 1 << 31         # device memory the game records of one trial may take

    def _trial_batched(self, num, game_init=None, depth=0, width=1, since_empty=6, limit_tile=0):
        """`num` games at once (r_learning.py:362-376 runs them one after the other), each a full reference `Game`: row, score,
        odometer, moves, tiles, starting_position — what `results[0].save_game` / show.py's replay need (game_logic.py:163-167
        appends every move, :118-121 every tile).  Everything happens on the device, the lanes' game logs switched on for all of
        them; `self.trial_seed = (seed, lane0)` fixes the lanes' RNG streams (default: away from the training lanes).
        depth 0, no tile limit: greedy TD steps with alpha = 0 (no records, no learning).
        depth > 0 or limit_tile: g2048_lookahead_steps — Game.trial_run with look-ahead (game_logic.py:150-183, 214-243) for all
        games in lock step, every game's expectimax tree expanded, evaluated and reduced in HBM (csrc/lookahead.hip); the chance
        nodes of a move are keyed by the lane's RNG state (rng.lookahead_draws)."""
        seed, lane0 = getattr(self, 'trial_seed', (self.seed + 77, 1 << 41))
        eng = Engine(num, seed=seed, lane0=lane0, share_table_of=self.engine)
        eng.set_auto_reset(False)
        if game_init is not None:
            eng.set_boards(np.repeat(np.asarray(game_init.row, np.uint8)[None], num, axis=0))
            eng.set_scores(np.full(num, game_init.score, np.int32))
        capacity = int(min(self.LOG_CAPACITY if depth == 0 else 4 * self.LOG_CAPACITY, max(2048, self.TRIAL_LOG_BYTES // (4 * num))))
        eng.log_enable(num, capacity)
        eng.stats_reset()
        searching = depth > 0 or bool(limit_tile)
        while eng.stats()['episodes'] < num:                   # every lane ends exactly once (auto-reset is off)
            if searching:
                eng.lookahead_steps(depth, width, since_empty, limit_tile, 64)
            else:
                eng.td_steps(0.0, 256)
        meta = eng.log_meta()
        games = []
        for lane in range(num):
            length, score, flags = int(meta[lane, 3]), int(meta[lane, 4]), int(meta[lane, 7])
            if flags & 4:                                      # longer than the record's capacity: the outcome without the moves
                game = Game(score=score, row=eng.log_final(lane, 0).astype(np.int32))
                game.starting_position, game.odometer = eng.log_game(lane, 0)[1].astype(np.int32), length
            else:
                game = self._game_from_log(eng, lane, 0, length, score)
                game.moves.pop()                               # trial_run ends without the -1 that episode() appends
            # (game_init: every game is `game_init.copy()` = Game(score, row), r_learning.py:363 / game_logic.py:69-70 — a fresh
            # record from that position: odometer 0, only the trial's own moves and tiles, starting_position = game_init.row;
            # which is what the device's record holds, its start being the board set above)
            games.append(game)
        eng.close()
        return games


Q_agent = QAgent          # the name used by the reference's README (README.md:62-65)

__all__ = [n for n in dir() if not n.startswith('_')]
