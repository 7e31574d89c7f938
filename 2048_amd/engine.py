"""Engine — a NumPy-facing handle on one g2048 device context (one GPU, one batch of lanes).

Thin by design: every method is one C-ABI call (include/g2048.h) plus buffer plumbing.  The
reference-shaped classes (`Game`, `QAgent`) in game_logic.py / r_learning.py are built on it.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import check

NUM_FEAT = {2: 24, 3: 52, 4: 17, 5: 21, 6: 33}              # QAgent.parameter_shape, r_learning.py:88


def _buf(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def table_slots(n, backend=None):
    """(geometry: the same in both libraries; `backend` says which one answers — an explicit Engine(backend='cpu') must not
    need the other library)"""
    return int(_lib.load(backend).g2048_table_slots(n))


def feature_layout(n, backend=None):
    F = NUM_FEAT[n]
    offs = np.zeros(F, np.int64)
    sizes = np.zeros(F, np.int64)
    check(_lib.load(backend).g2048_feature_layout(n, _buf(offs), _buf(sizes)))
    return offs, sizes


class Engine:
    def __init__(self, batch, n=0, seed=2048, lane0=0, device=0, share_table_of=None, backend=None):
        """share_table_of: another Engine whose weight table this one uses (g2048_create_shared).
        backend: 'hip' (lib2048_hip.so, the default) or 'cpu' (lib2048_cpu.so, csrc/cpu_ref.cpp) — None takes G2048_BACKEND
        from the environment; an explicit choice either way, never a fallback."""
        if share_table_of is not None:
            n, device, backend = share_table_of.n, share_table_of.device, share_table_of.backend
        self.backend = backend or _lib.default_backend()
        self.lib = _lib.load(self.backend)
        self.batch, self.n, self.seed, self.lane0, self.device = int(batch), int(n), int(seed), int(lane0), int(device)
        self.num_feat = NUM_FEAT.get(self.n, 0)
        self.slots = table_slots(self.n, self.backend) if self.n else 0
        self.parent = share_table_of                      # keeps the table's owner alive
        ctx = ctypes.c_void_p()
        if share_table_of is None:
            check(self.lib.g2048_create(self.device, self.batch, self.n, self.seed & (2 ** 64 - 1), self.lane0, ctypes.byref(ctx)), None, self.lib)
        else:
            check(self.lib.g2048_create_shared(share_table_of.ctx, self.batch, self.seed & (2 ** 64 - 1), self.lane0, ctypes.byref(ctx)), None, self.lib)
        self.ctx = ctx

    def close(self):
        if getattr(self, 'ctx', None):
            self.lib.g2048_destroy(self.ctx)
            self.ctx = None

    __del__ = close

    def _c(self, status):
        check(status, self.ctx, self.lib)

    # ---- lane state
    def set_boards(self, boards, clear_carry=True):
        b = np.ascontiguousarray(np.asarray(boards).reshape(self.batch, 16), np.uint8)
        self._c(self.lib.g2048_set_boards(self.ctx, _buf(b)))
        if clear_carry:
            self._c(self.lib.g2048_clear_carry(self.ctx))

    def get_boards(self):
        b = np.empty((self.batch, 4, 4), np.uint8)
        self._c(self.lib.g2048_get_boards(self.ctx, _buf(b)))
        return b

    def set_scores(self, scores):
        s = np.ascontiguousarray(scores, np.int32).reshape(self.batch)
        self._c(self.lib.g2048_set_scores(self.ctx, _buf(s)))

    def get_scores(self):
        s = np.empty(self.batch, np.int32)
        self._c(self.lib.g2048_get_scores(self.ctx, _buf(s)))
        return s

    def set_rng(self, state):
        s = np.ascontiguousarray(state, np.uint64).reshape(self.batch, 2)
        self._c(self.lib.g2048_set_rng(self.ctx, _buf(s)))

    def get_rng(self):
        s = np.empty((self.batch, 2), np.uint64)
        self._c(self.lib.g2048_get_rng(self.ctx, _buf(s)))
        return s

    def get_carry(self):
        prev = np.empty((self.batch, 4, 4), np.uint8)
        label = np.empty(self.batch, np.float32)
        flags = np.empty(self.batch, np.uint8)
        self._c(self.lib.g2048_get_carry(self.ctx, _buf(prev), _buf(label), _buf(flags)))
        return prev, label, flags

    def clear_carry(self):
        self._c(self.lib.g2048_clear_carry(self.ctx))

    def reset(self):
        self._c(self.lib.g2048_reset(self.ctx))

    def set_auto_reset(self, on):
        self._c(self.lib.g2048_set_auto_reset(self.ctx, int(bool(on))))

    def sync(self):
        self._c(self.lib.g2048_sync(self.ctx))

    # ---- environment
    def move_all(self):
        after = np.empty((self.batch, 4, 4, 4), np.uint8)
        reward = np.empty((self.batch, 4), np.int32)
        changed = np.empty(self.batch, np.uint8)
        self._c(self.lib.g2048_move_all(self.ctx, _buf(after), _buf(reward), _buf(changed)))
        return after, reward, changed

    def apply_moves(self, dirs):
        d = np.ascontiguousarray(dirs, np.uint8).reshape(self.batch)
        moved = np.empty(self.batch, np.uint8)
        self._c(self.lib.g2048_apply_moves(self.ctx, _buf(d), _buf(moved)))
        return moved.astype(bool)

    def terminal(self):
        over, ne, npairs = (np.empty(self.batch, np.uint8) for _ in range(3))
        self._c(self.lib.g2048_terminal(self.ctx, _buf(over), _buf(ne), _buf(npairs)))
        return over.astype(bool), ne, npairs

    def spawn(self):
        r10 = np.empty(self.batch, np.uint8)
        k = np.empty(self.batch, np.uint8)
        self._c(self.lib.g2048_spawn(self.ctx, _buf(r10), _buf(k)))
        return r10, k

    def spawn_injected(self, r10, k):
        r = np.ascontiguousarray(r10, np.uint8).reshape(self.batch)
        kk = np.ascontiguousarray(k, np.uint8).reshape(self.batch)
        self._c(self.lib.g2048_spawn_injected(self.ctx, _buf(r), _buf(kk)))

    def boards_move_all(self, boards):
        """Stateless: (after[n,4,4,4], reward[n,4], changed[n]) of any number of boards."""
        b = np.ascontiguousarray(np.asarray(boards).reshape(-1, 16), np.uint8)
        n = len(b)
        after = np.empty((n, 4, 4, 4), np.uint8)
        reward = np.empty((n, 4), np.int32)
        changed = np.empty(n, np.uint8)
        self._c(self.lib.g2048_boards_move_all(self.ctx, _buf(b), n, _buf(after), _buf(reward), _buf(changed)))
        return after, reward, changed

    def boards_evaluate(self, boards):
        """Stateless: V(board) for any number of boards with this context's table."""
        b = np.ascontiguousarray(np.asarray(boards).reshape(-1, 16), np.uint8)
        v = np.empty(len(b), np.float32)
        self._c(self.lib.g2048_boards_evaluate(self.ctx, _buf(b), len(b), _buf(v)))
        return v

    def boards_look_forward(self, boards, depth, width, since_empty, salt=None):
        """Stateless: V_depth(board) of Game.look_forward (game_logic.py:214-243) for any number of boards, every tree expanded and
        reduced on the device; salt uint64[n, 2] keys the chance nodes (rng.lookahead_draws), None = zeros."""
        b = np.ascontiguousarray(np.asarray(boards).reshape(-1, 16), np.uint8)
        v = np.empty(len(b), np.float32)
        s = None if salt is None else np.ascontiguousarray(salt, np.uint64).reshape(len(b), 2)
        self._c(self.lib.g2048_boards_look_forward(self.ctx, _buf(b), len(b), int(depth), int(width), int(since_empty), _buf(s) if s is not None else None, _buf(v)))
        return v

    def lookahead_steps(self, depth, width, since_empty, limit_tile=0, nsteps=1):
        """nsteps moves of Game.trial_run with look-ahead for every live lane (choice, move, new tile, end test), on the device."""
        self._c(self.lib.g2048_lookahead_steps(self.ctx, int(depth), int(width), int(since_empty), int(limit_tile), int(nsteps)))

    def step_random(self, nsteps):
        self._c(self.lib.g2048_step_random(self.ctx, int(nsteps)))

    # ---- features / value
    def features(self):
        out = np.empty((self.batch, self.num_feat), np.int32)
        self._c(self.lib.g2048_features(self.ctx, _buf(out)))
        return out

    def set_weights(self, w):
        w = np.ascontiguousarray(w, np.float32).reshape(-1)
        self._c(self.lib.g2048_weights_set(self.ctx, _buf(w), w.size))

    def get_weights(self):
        w = np.empty(self.slots, np.float32)
        self._c(self.lib.g2048_weights_get(self.ctx, _buf(w), w.size))
        return w

    def init_weights(self, seed=0, scale=0.01):
        self._c(self.lib.g2048_weights_init(self.ctx, int(seed) & (2 ** 64 - 1), float(scale)))

    def evaluate(self):
        v = np.empty(self.batch, np.float32)
        self._c(self.lib.g2048_evaluate(self.ctx, _buf(v)))
        return v

    def eval_select(self, want_all=False):
        v = np.empty(self.batch, np.float32)
        a = np.empty(self.batch, np.uint8)
        v4 = np.empty((self.batch, 4), np.float32) if want_all else None
        self._c(self.lib.g2048_eval_select(self.ctx, _buf(v), _buf(a), _buf(v4) if want_all else None))
        return (v, a, v4) if want_all else (v, a)

    def eval_select_device_only(self):
        """Run the greedy-choice kernel without copying anything back (timing)."""
        self._c(self.lib.g2048_eval_select(self.ctx, None, None, None))

    # ---- learning
    def update(self, states, dw):
        s = np.ascontiguousarray(np.asarray(states).reshape(-1, 16), np.uint8)
        d = np.ascontiguousarray(dw, np.float32).reshape(-1)
        assert len(s) == len(d)
        self._c(self.lib.g2048_update(self.ctx, _buf(s), _buf(d), len(d)))

    def td_steps(self, alpha, nsteps=1):
        self._c(self.lib.g2048_td_steps(self.ctx, float(alpha), int(nsteps)))

    def last_move(self):
        """uint16[B], see g2048_get_last_move: direction | moved<<2 | cell<<4 | tile<<8 | spawned<<10 | ended<<11."""
        out = np.empty(self.batch, np.uint16)
        self._c(self.lib.g2048_get_last_move(self.ctx, _buf(out)))
        return out

    def log_enable(self, lanes, capacity):
        """Keep the moves / tiles of every game of the first `lanes` lanes (up to `capacity` moves each)."""
        self._log_geometry = (int(lanes), int(capacity))
        self._c(self.lib.g2048_log_enable(self.ctx, int(lanes), int(capacity)))

    def log_meta(self):
        lanes, _ = self._log_geometry
        out = np.empty((lanes, 8), np.uint32)
        self._c(self.lib.g2048_log_meta(self.ctx, _buf(out)))
        return out

    def log_game(self, lane, slot):
        """(moves uint16[capacity], start uint8[4,4]) of one recorded game."""
        _, cap = self._log_geometry
        moves = np.empty(cap, np.uint16)
        start = np.empty((4, 4), np.uint8)
        self._c(self.lib.g2048_log_game(self.ctx, int(lane), int(slot), _buf(moves), _buf(start)))
        return moves, start

    def log_final(self, lane, slot):
        """uint8[4,4]: the board a finished recorded game ended on."""
        board = np.empty((4, 4), np.uint8)
        self._c(self.lib.g2048_log_final(self.ctx, int(lane), int(slot), _buf(board)))
        return board

    def set_lane_sort(self, every):
        """Re-order the lanes by their big-tile pattern every `every` TD steps (0 = never); invisible to the host."""
        self._c(self.lib.g2048_set_lane_sort(self.ctx, int(every)))

    def debug_lane_order(self):
        """(perm, keys) of the lane re-order run on the current boards (test hook; the order itself is not changed)."""
        perm, keys = np.zeros(self.batch, np.uint32), np.zeros(self.batch, np.uint16)
        self._c(self.lib.g2048_debug_lane_order(self.ctx, _buf(perm), _buf(keys)))
        return perm, keys

    def set_update_mode(self, mode):
        """1 = LDS-owner update kernel (default), 0 = global fp32 atomics."""
        self._c(self.lib.g2048_set_update_mode(self.ctx, int(mode)))

    def set_update_rule(self, rule):
        """0 = add every dw (the reference's update), 1 = per-slot mean of the step's dw."""
        self._c(self.lib.g2048_set_update_rule(self.ctx, int(rule)))

    def stats(self):
        st = _lib.Stats()
        self._c(self.lib.g2048_stats_get(self.ctx, ctypes.byref(st)))
        return dict(episodes=st.episodes, moves=st.moves, score_sum=st.score_sum, best_score=st.best_score,
                    max_tile=list(st.max_tile), overflow16=st.overflow16, nonfinite=st.nonfinite, valid_dirs=st.valid_dirs)

    def stats_reset(self):
        self._c(self.lib.g2048_stats_reset(self.ctx))

    # ---- timing (HIP events on the context's own stream)
    def timer_start(self):
        self._c(self.lib.g2048_timer_start(self.ctx))

    def timer_stop(self):
        ms = ctypes.c_float()
        self._c(self.lib.g2048_timer_stop(self.ctx, ctypes.byref(ms)))
        return ms.value

    # ---- multi-GPU plumbing
    def weights_ptr(self):
        p, n = ctypes.c_void_p(), ctypes.c_int64()
        self._c(self.lib.g2048_weights_device_ptr(self.ctx, ctypes.byref(p), ctypes.byref(n)))
        return p.value, n.value

    def delta_begin(self):
        self._c(self.lib.g2048_delta_begin(self.ctx))

    def delta_extract(self, dst_ptr=None):
        """delta = W - W0 into the device buffer at dst_ptr (e.g. tensor.data_ptr()) or the context's own."""
        self._c(self.lib.g2048_delta_extract(self.ctx, ctypes.c_void_p(dst_ptr)))

    def delta_apply(self, src_ptr=None):
        self._c(self.lib.g2048_delta_apply(self.ctx, ctypes.c_void_p(src_ptr)))

    def delta_pack_touched(self, pack_ptr):
        """pack fp32[2 * slots] = [delta | 1.0 where this rank moved the slot] (per-slot mean rule across ranks)."""
        self._c(self.lib.g2048_delta_pack_touched(self.ctx, ctypes.c_void_p(pack_ptr)))

    def delta_apply_mean(self, pack_ptr):
        """W = W0 + delta_sum / max(1, touched_sum) from the all-reduced pack."""
        self._c(self.lib.g2048_delta_apply_mean(self.ctx, ctypes.c_void_p(pack_ptr)))

    # native RCCL path (include/g2048.h, multi-GPU)
    @staticmethod
    def comm_unique_id(backend=None):
        lib = _lib.load(backend)
        buf = (ctypes.c_uint8 * _lib.COMM_ID_BYTES)()
        check(lib.g2048_comm_unique_id(buf))
        return bytes(buf)

    def comm_init(self, rank, nranks, unique_id):
        assert len(unique_id) == _lib.COMM_ID_BYTES
        buf = (ctypes.c_uint8 * _lib.COMM_ID_BYTES).from_buffer_copy(unique_id)
        self._c(self.lib.g2048_comm_init(self.ctx, int(rank), int(nranks), buf))

    def comm_destroy(self):
        self._c(self.lib.g2048_comm_destroy(self.ctx))

    def comm_info(self):
        """(rank, nranks) as the RCCL communicator itself reports them; (0, 1) without one."""
        r, n = ctypes.c_int(), ctypes.c_int()
        self._c(self.lib.g2048_comm_info(self.ctx, ctypes.byref(r), ctypes.byref(n)))
        return r.value, n.value

    def allreduce_deltas(self):
        """End of an epoch on the context's stream: RCCL sum all-reduce of the accumulated delta, W = W0 + result."""
        self._c(self.lib.g2048_allreduce_deltas(self.ctx))

    def allreduce_f64(self, values, op='sum'):
        v = np.ascontiguousarray(values, np.float64).copy()
        self._c(self.lib.g2048_allreduce_f64(self.ctx, _buf(v), v.size, 1 if op == 'max' else 0))
        return v

    def delta_ptr(self):
        p = ctypes.c_void_p()
        self._c(self.lib.g2048_delta_device_ptr(self.ctx, ctypes.byref(p)))
        return p.value

    def debug_owner_plan(self):
        """[(variant, chunk, part, nparts, start_clock, end_clock)] of the last LDS-owner launch (100 MHz clock)."""
        out = np.zeros((1024, 6), np.uint64)
        n = ctypes.c_uint32(0)
        self._c(self.lib.g2048_debug_owner_plan(self.ctx, _buf(out), 1024, ctypes.byref(n)))
        return out[:n.value]

    def td_steps_kernel_ms(self, alpha, nsteps):
        """ms per launch of (k_td_play, k_td_update_owner, k_td_update_tail, k_apply_*), HIP events on the context's stream."""
        out = np.zeros(4, np.float32)
        self._c(self.lib.g2048_td_steps_kernel_ms(self.ctx, float(alpha), int(nsteps), _buf(out)))
        return [float(x) for x in out]

    def td_steps_profiled(self, alpha, nsteps):
        """(ms per k_td_play launch, ms per k_td_update launch), HIP events on the context's stream."""
        a, b = ctypes.c_float(), ctypes.c_float()
        self._c(self.lib.g2048_td_steps_profiled(self.ctx, float(alpha), int(nsteps), ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value
