"""lib2048_cpu.so (2048_amd/csrc/cpu_ref.cpp): the same C ABI on the host, from the same integer headers as the HIP kernels.

These are the GPU suite's own parity tests — imported from tests/test_gpu_parity.py / test_gpu_surface.py and run unchanged
against the CPU backend (G2048_BACKEND=cpu for the duration of each test) — at sizes a CPU finishes in seconds.  They see
the library only through ctypes, exactly as on the GPU box: every integer result bit-exact against the reference-generated
fixtures and the oracle, values and table updates within the stated fp32 tolerances, the reference's own episode(), trial()
and look_forward() outputs reproduced.  (The whole GPU suite minus its device-only diagnostics passes on this backend:
99 of 108 tests in 24 min with 8 threads; the selection below keeps `-m "not gpu"` at a few minutes.)

The backend is chosen EXPLICITLY: nothing falls back to it (tests/test_cpu_host.py checks that a missing GPU is an error
under the default backend).
"""
import ctypes
import importlib
import os

import numpy as np
import pytest

from tests import test_gpu_parity as gp
from tests import test_gpu_surface as gs

pkg = importlib.import_module('2048_amd')
_lib = importlib.import_module('2048_amd._lib')


@pytest.fixture(autouse=True)
def cpu_backend(monkeypatch):
    monkeypatch.setenv('G2048_BACKEND', 'cpu')
    monkeypatch.setenv('G2048_CPU_THREADS', '4')
    yield


@pytest.fixture(scope='module')
def api():
    os.environ['G2048_BACKEND'] = 'cpu'         # (module-scoped: the alias package builds its lazy 1-lane engines on first use)
    try:
        import game2048.r_learning as rl
        yield rl
    finally:
        os.environ.pop('G2048_BACKEND', None)


def test_cpu_library_exports_the_whole_abi():
    lib = ctypes.CDLL(_lib.CPU_LIB_PATH)
    for name in _lib.SIGNATURES:
        assert hasattr(lib, name), f'{name} is declared in include/g2048.h but not exported by lib2048_cpu.so'
    assert _lib.load('cpu').g2048_abi_version() == 2
    eng = pkg.Engine(4, n=2)
    assert eng.backend == 'cpu' and eng.lib is _lib.load('cpu')
    eng.close()


# ---- environment: a-1 .. a-5
def test_move_table(golden):
    gp.test_move_table_all_rows_all_directions(golden)


def test_moves_terminal_spawn(golden):
    gp.test_golden_moves_terminal_spawn(golden)
    gp.test_full_boards_do_not_spawn_and_rng_untouched()


def test_rng_streams():
    gp.test_rng_stream_new_games_and_exported_draws()


def test_step_random_vs_oracle():
    gp.test_step_random_config2(777, 400)
    gp.test_step_random_without_auto_reset_runs_every_game_to_the_end()


# ---- features, value, greedy choice, update: a-6 .. a-12
@pytest.mark.parametrize('n', [2, 3, 4, 5, 6])
def test_features_value_select_update(golden, n):
    gp.test_features_golden(golden, n)
    gp.test_value_select_update_golden(golden, n)


# ---- the TD(0) loop: a-11 .. a-13
@pytest.mark.parametrize('n', [2, 3, 4, 5, 6])
def test_reference_episode_traces(golden, n):
    gp.test_td_single_lane_reproduces_reference_episode(golden, n, 1)
    if n <= 4:
        gp.test_td_single_lane_reproduces_reference_episode(golden, n, 0)


@pytest.mark.parametrize('n', [2, 3, 4, 5])
def test_batched_td_vs_oracle(n):
    gp.test_td_steps_batch_vs_oracle(n, 1)
    gp.test_td_mean_rule_vs_oracle(n)


def test_batched_td_ragged_and_to_the_end():
    gp.test_td_ragged_lane_counts(3, 5, 'sum')
    gp.test_td_ragged_lane_counts(1000, 4, 'mean')
    gp.test_td_ragged_lane_counts(777, 3, 'sum')
    gp.test_td_batch_until_all_games_end(1)


def test_whole_game_fp32_model_and_switches():
    gp.test_td_whole_game_fp32_model_bit_exact(1)
    gp.test_td_whole_game_fp32_model_bit_exact(0)
    gp.test_td_rule_and_mode_switches_mid_run(3)


def test_shared_tables_last_move_delta_protocol_stats():
    gp.test_shared_table_contexts_and_last_move()
    gp.test_weight_delta_protocol_on_device()


# ---- the reference-shaped surface on this backend: what show.py calls on a box without a GPU
def test_surface_game_and_agent(api, golden, tmp_path):
    gs.test_game_moves_terminal_and_table(api, golden)
    gs.test_game_play_and_replay(api)
    gs.test_agent_evaluate_update_features(api, golden, 2)
    gs.test_agent_evaluate_update_features(api, golden, 4)
    gs.test_agent_episode_returns_a_replayable_game(api)
    gs.test_agent_pickle_round_trip(api, tmp_path)


def test_surface_trial_reproduces_the_reference_trial(api, golden, tmp_path):
    gs.test_trial_reproduces_the_reference_trial(api, golden, tmp_path)


def test_surface_look_forward_and_reference_pickles(api, golden, tmp_path, monkeypatch):
    gs.test_look_forward_matches_reference_fixture(api, golden)
    gs.test_reference_written_pickles_load(api, golden, tmp_path, monkeypatch)
    gs.test_device_game_records(api)


def test_threads_do_not_change_the_games(monkeypatch):
    """1 thread and 4 threads play the same games (lane logic is per lane; the table sums are float64 per slot, so the tables
    agree to the last fp32 rounding of a sum whose order differs)."""
    out = []
    for threads in ('1', '4'):
        monkeypatch.setenv('G2048_CPU_THREADS', threads)
        eng = pkg.Engine(3000, n=4, seed=5)
        eng.init_weights(seed=3, scale=0.01)
        eng.td_steps(0.25 * 17 / (8 * 3000), 30)
        out.append((eng.get_boards(), eng.get_scores(), eng.get_rng(), eng.get_weights(), eng.stats()))
        eng.close()
    a, b = out
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and a[4] == b[4]
    assert np.abs(a[3] - b[3]).max() <= 2e-7 * max(1.0, np.abs(a[3]).max())
