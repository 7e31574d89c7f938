"""ctypes binding of include/g2048.h.

Two libraries implement the ABI: lib2048_hip.so (the product: hand-written HIP for gfx950) and lib2048_cpu.so (csrc/cpu_ref.cpp:
the same entry points in scalar C++ from the same integer headers, for boxes without a GPU and as a CPU baseline).  Which one
a process uses is an EXPLICIT choice — `Engine(backend='cpu')` or G2048_BACKEND=cpu in the environment; the default is 'hip'
and there is no fallback from one to the other: if lib2048_hip.so has not been built, loading fails loudly; if it is built
but no GPU is visible, `g2048_create` returns G2048_ERR_NODEV and `check` raises.  Build with
`python -c "import __graft_entry__ as g; g.build()"` or `make -C 2048_amd/csrc`.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_uint8, c_uint32, c_uint64, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('G2048_LIB') or os.path.join(HERE, 'lib2048_hip.so')        # (G2048_LIB: experiment builds)
CPU_LIB_PATH = os.path.join(HERE, 'lib2048_cpu.so')
BACKENDS = ('hip', 'cpu')


def default_backend():
    """'hip' unless G2048_BACKEND says otherwise (read at every call: a test may set it for one engine)."""
    b = os.environ.get('G2048_BACKEND', 'hip')
    if b not in BACKENDS:
        raise ValueError(f'G2048_BACKEND={b!r}: expected one of {BACKENDS}')
    return b

OK, ERR_ARG, ERR_HIP, ERR_NOMEM, ERR_STATE, ERR_NODEV, ERR_COMM = 0, -1, -2, -3, -4, -5, -6
COMM_ID_BYTES = 128
LANE_HAS_PREV, LANE_DONE = 1, 2


class G2048Error(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f'g2048 status {status}: {message}')
        self.status = status


class Stats(ctypes.Structure):
    _fields_ = [('episodes', c_uint64), ('moves', c_uint64), ('score_sum', c_uint64), ('best_score', c_uint64),
                ('max_tile', c_uint64 * 20), ('overflow16', c_uint64), ('nonfinite', c_uint64), ('valid_dirs', c_uint64)]


_P = c_void_p          # opaque context / generic buffers (numpy arrays are passed by address)

# name -> (restype, argtypes); every entry point declared in include/g2048.h
SIGNATURES = {
    'g2048_abi_version': (c_int, []),
    'g2048_strerror': (c_char_p, [c_int]),
    'g2048_device_count': (c_int, [POINTER(c_int)]),
    'g2048_num_feat': (c_int, [c_int]),
    'g2048_table_slots': (c_int64, [c_int]),
    'g2048_feature_layout': (c_int, [c_int, _P, _P]),
    'g2048_create': (c_int, [c_int, c_uint32, c_int, c_uint64, c_uint64, POINTER(_P)]),
    'g2048_create_shared': (c_int, [_P, c_uint32, c_uint64, c_uint64, POINTER(_P)]),
    'g2048_destroy': (c_int, [_P]),
    'g2048_last_error': (c_char_p, [_P]),
    'g2048_sync': (c_int, [_P]),
    'g2048_timer_start': (c_int, [_P]),
    'g2048_timer_stop': (c_int, [_P, POINTER(c_float)]),
    'g2048_set_boards': (c_int, [_P, _P]),
    'g2048_get_boards': (c_int, [_P, _P]),
    'g2048_set_scores': (c_int, [_P, _P]),
    'g2048_get_scores': (c_int, [_P, _P]),
    'g2048_set_rng': (c_int, [_P, _P]),
    'g2048_get_rng': (c_int, [_P, _P]),
    'g2048_get_carry': (c_int, [_P, _P, _P, _P]),
    'g2048_clear_carry': (c_int, [_P]),
    'g2048_reset': (c_int, [_P]),
    'g2048_set_auto_reset': (c_int, [_P, c_int]),
    'g2048_move_all': (c_int, [_P, _P, _P, _P]),
    'g2048_apply_moves': (c_int, [_P, _P, _P]),
    'g2048_terminal': (c_int, [_P, _P, _P, _P]),
    'g2048_spawn': (c_int, [_P, _P, _P]),
    'g2048_spawn_injected': (c_int, [_P, _P, _P]),
    'g2048_boards_move_all': (c_int, [_P, _P, c_int64, _P, _P, _P]),
    'g2048_boards_evaluate': (c_int, [_P, _P, c_int64, _P]),
    'g2048_boards_look_forward': (c_int, [_P, _P, c_int64, c_int, c_int, c_int, _P, _P]),
    'g2048_lookahead_steps': (c_int, [_P, c_int, c_int, c_int, c_int, c_uint32]),
    'g2048_step_random': (c_int, [_P, c_uint32]),
    'g2048_features': (c_int, [_P, _P]),
    'g2048_weights_set': (c_int, [_P, _P, c_int64]),
    'g2048_weights_get': (c_int, [_P, _P, c_int64]),
    'g2048_weights_init': (c_int, [_P, c_uint64, c_float]),
    'g2048_evaluate': (c_int, [_P, _P]),
    'g2048_eval_select': (c_int, [_P, _P, _P, _P]),
    'g2048_update': (c_int, [_P, _P, _P, c_int64]),
    'g2048_td_steps': (c_int, [_P, c_float, c_uint32]),
    'g2048_set_lane_sort': (c_int, [_P, c_uint32]),
    'g2048_debug_lane_order': (c_int, [_P, _P, _P]),
    'g2048_set_update_mode': (c_int, [_P, c_int]),
    'g2048_set_update_rule': (c_int, [_P, c_int]),
    'g2048_get_last_move': (c_int, [_P, _P]),
    'g2048_log_enable': (c_int, [_P, c_uint32, c_uint32]),
    'g2048_log_meta': (c_int, [_P, _P]),
    'g2048_log_game': (c_int, [_P, c_uint32, c_uint32, _P, _P]),
    'g2048_log_final': (c_int, [_P, c_uint32, c_uint32, _P]),
    'g2048_stats_get': (c_int, [_P, POINTER(Stats)]),
    'g2048_stats_reset': (c_int, [_P]),
    'g2048_weights_device_ptr': (c_int, [_P, POINTER(_P), POINTER(c_int64)]),
    'g2048_delta_begin': (c_int, [_P]),
    'g2048_delta_extract': (c_int, [_P, _P]),
    'g2048_delta_apply': (c_int, [_P, _P]),
    'g2048_delta_device_ptr': (c_int, [_P, POINTER(_P)]),
    'g2048_td_steps_profiled': (c_int, [_P, c_float, c_uint32, POINTER(c_float), POINTER(c_float)]),
    'g2048_debug_owner_plan': (c_int, [_P, _P, c_uint32, POINTER(c_uint32)]),
    'g2048_stream_handle': (c_int, [_P, POINTER(_P)]),
    'g2048_td_steps_kernel_ms': (c_int, [_P, c_float, c_uint32, _P]),
    'g2048_delta_pack_touched': (c_int, [_P, _P]),
    'g2048_delta_apply_mean': (c_int, [_P, _P]),
    'g2048_comm_unique_id': (c_int, [_P]),
    'g2048_comm_init': (c_int, [_P, c_int, c_int, _P]),
    'g2048_comm_destroy': (c_int, [_P]),
    'g2048_comm_info': (c_int, [_P, POINTER(c_int), POINTER(c_int)]),
    'g2048_allreduce_deltas': (c_int, [_P]),
    'g2048_allreduce_f64': (c_int, [_P, _P, c_int, c_int]),
}

_libs = {}


def load(backend=None):
    """Load the library of `backend` ('hip' | 'cpu' | None = default_backend()), once.  Raises if it is missing — there is no
    fallback between the two."""
    backend = backend or default_backend()
    if backend not in _libs:
        path = LIB_PATH if backend == 'hip' else CPU_LIB_PATH
        if not os.path.exists(path):
            raise G2048Error(ERR_NODEV, f'{path} is not built; run __graft_entry__.build() '
                                        + ('(hipcc --offload-arch=gfx950).  There is no CPU fallback.' if backend == 'hip' else '(g++ -fopenmp csrc/cpu_ref.cpp).'))
        lib = ctypes.CDLL(path)
        experiment = backend == 'hip' and bool(os.environ.get('G2048_LIB'))      # (a timing build of tools/exp, possibly of an older ABI)
        for name, (res, args) in SIGNATURES.items():
            if experiment and not hasattr(lib, name):
                continue
            fn = getattr(lib, name)          # AttributeError here = header and library disagree
            fn.restype = res
            fn.argtypes = args
        if lib.g2048_abi_version() != 3 and not experiment:
            raise G2048Error(ERR_STATE, 'ABI version mismatch')
        _libs[backend] = lib
    return _libs[backend]


def check(status, ctx=None, lib=None):
    if status != OK:
        lib = lib or load()
        msg = lib.g2048_strerror(status).decode()
        if ctx:
            detail = lib.g2048_last_error(ctx).decode()
            if detail:
                msg += ': ' + detail
        raise G2048Error(status, msg)
