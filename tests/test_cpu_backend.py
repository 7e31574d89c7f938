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
    assert _lib.load('cpu').g2048_abi_version() == 3
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


def test_lookahead_steps_depth_zero():
    gp.test_lookahead_steps_at_depth_zero_are_the_greedy_steps(4)


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
    gs.test_trial_with_game_init_reproduces_the_reference(api, golden)


def test_surface_look_forward_and_reference_pickles(api, golden, tmp_path, monkeypatch):
    gs.test_look_forward_matches_reference_fixture(api, golden)
    gs.test_device_look_forward_matches_the_reference(golden)
    gs.test_game_find_best_move_with_depth_goes_through_the_device_look_ahead(api, golden, monkeypatch)
    gs.test_trial_with_lookahead_reproduces_the_reference_trial(api, golden)
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


def test_qagent_train_run_two_ranks_on_the_cpu_backend(tmp_path):
    """QAgent.train_run as a two-rank torch.distributed job (gloo), each rank a real Engine on the CPU backend: the ranks play
    different lane shards, exchange their accumulated weight deltas every epoch through parallel.DeltaSync (the host-pointer
    form of g2048_delta_*), stay identical replicas with identical schedules, agree on the job's best recorded game, and
    rank 0 reports for the job.  The same script is the 8-GPU training entry point (tools/train_multi.py)."""
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    dump = str(tmp_path / 'rank')
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   G2048_BACKEND='cpu', G2048_CPU_THREADS='2')
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, 'tools', 'train_multi.py'), '--n', '2', '--batch', '512', '--episodes', '4000',
                                       '--epoch', '16', '--backend', 'gloo', '--comm', 'torch', '--dump', dump], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs[0][1][-1500:] + outs[1][1][-1500:]
    a, b = (np.load(f'{dump}.{r}.npz') for r in range(2))
    assert int(a['step']) == int(b['step']) >= 4000 and float(a['alpha']) == float(b['alpha']) and int(a['top_tile']) == int(b['top_tile'])
    assert np.array_equal(a['history'], b['history']) and len(a['history']) > 5
    assert np.array_equal(a['w_head'], b['w_head']) and np.array_equal(a['w_tail'], b['w_tail']) and float(a['wsum']) == float(b['wsum'])
    assert int(a['reduces']) > 3 and str(a['sync']) == 'DeltaSync'
    assert int(a['top_game_score']) == int(b['top_game_score']) > 0          # one best game for the whole job
    assert int(a['top_game_score']) <= int(a['top_score'])
    assert 'training session started' in outs[0][0] and 'on each of 2 GPUs' in outs[0][0] and 'training session started' not in outs[1][0]


def _bench_cpu(*extra, timeout=300, gpus=2, n=4, batch=2048, threads=2):
    import json
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, G2048_BACKEND='cpu', G2048_CPU_THREADS=str(threads))
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    t0 = time.monotonic()
    res = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', str(gpus), '--backend', 'gloo', '--batch', str(batch), '--n-tuple', str(n), '--steps', '6',
                          '--warmup', '2', '--condition', '4', '--epoch', '3', '--repeats', '1', '--no-cpu-baseline', '--trained-steps', '0', *extra],
                         capture_output=True, text=True, timeout=timeout, env=env)
    line = json.loads(res.stdout.strip().splitlines()[-1]) if res.returncode == 0 and res.stdout.strip() else None
    return res, line, time.monotonic() - t0


def test_bench_two_ranks_whole_path_on_the_cpu_backend():
    """bench.py --gpus 2 end to end with real engines (CPU backend), gloo as transport: the self-launcher, lane sharding by rank,
    the epoch loop, the native-path set-up protocol (no RCCL here: EVERY rank falls back to torch.distributed together), the
    contract's JSON line with its comm block."""
    res, line, _ = _bench_cpu('--comm', 'native')
    assert res.returncode == 0, res.stderr[-2000:]
    assert line['n_gpus'] == 2 and line['steps'] == 6 and line['scaling'] == 'weak' and line['value'] > 0
    assert line['comm']['nranks_seen'] == 2 and line['comm']['exchanges_per_timed_region'] == 2 and line['comm']['allreduce_plus_apply_ms'] > 0
    assert 'torch.distributed all_reduce (gloo)' in line['comm']['kind'] and 'native RCCL path unavailable' in res.stderr
    assert line['roofline']['frac'] <= 1.0 and 'invalid' not in line['roofline']


def test_bench_rank_killed_mid_run_on_the_cpu_backend():
    """One rank dies between two exchanges; the survivor would sit in the next all-reduce.  The parent ends the job within
    seconds, exits non-zero and prints no result line."""
    res, line, dt = _bench_cpu('--comm', 'torch', '--fault-inject', 'exit@1:run')
    assert res.returncode != 0 and line is None and dt < 120
    assert 'rank 1 exited with code 3' in res.stderr and res.stdout.strip() == ''


# ---- BASELINE config 5 (8 x MI355X, 6-tuple table, per-epoch all-reduce of the weight deltas): its first run on hardware is the
# driver's, so the exact command is rehearsed here with eight real engines on the CPU backend — everything but RCCL itself

def test_bench_config5_eight_ranks_n6_rehearsal():
    """`bench.py --gpus 8 --n-tuple 6` through the self-launcher with eight ranks (gloo as transport, reduced --batch): the
    eight-rank rendezvous, lane0 = rank x B up to rank 7, the 382.65 MB exchange through run_epochs, the native path's set-up
    protocol failing on EVERY rank together, the line's comm block with the evidence of what ran."""
    res, line, _ = _bench_cpu('--comm', 'native', gpus=8, n=6, batch=256, threads=1, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    assert line['n_gpus'] == 8 and line['steps'] == 6 and line['scaling'] == 'weak' and line['value'] > 0
    assert 'BASELINE config 5' in line['config']['workload'] and line['config']['n_tuple'] == 6
    comm = line['comm']
    assert comm['nranks_seen'] == 8 and comm['nranks_expected'] == 8 and comm['payload_bytes'] == 95662848 * 4
    assert comm['lane0_per_rank'] == [r * 256 for r in range(8)]            # eight distinct shards, rank 7 included
    assert comm['replicas_identical'] and len({tuple(c) for c in comm['table_checksums_per_rank']}) == 1
    assert comm['exchanges_per_timed_region'] == 2 and comm['allreduce_plus_apply_ms'] > 0
    assert res.stderr.count('native RCCL path unavailable') == 8            # all eight fell back together


def test_bench_eight_ranks_rank_killed_and_deadline():
    """The launcher's failure handling at N = 8: rank 5 dies between two exchanges while seven survivors wait in the all-reduce
    (the parent ends the job, names the rank, prints no line); two ranks hang at start-up and --deadline ends the run."""
    res, line, dt = _bench_cpu('--comm', 'torch', '--fault-inject', 'exit@5:run', gpus=8, batch=512, threads=1)
    assert res.returncode != 0 and line is None and dt < 150
    assert 'rank 5 exited with code 3' in res.stderr and res.stdout.strip() == ''
    res, line, dt = _bench_cpu('--comm', 'torch', '--fault-inject', 'hang@3:init,hang@6:init', '--deadline', '25', gpus=8, batch=512, threads=1)
    assert res.returncode != 0 and line is None and 20 < dt < 120
    assert '--deadline 25 s passed' in res.stderr and res.stdout.strip() == ''


def test_bench_side_workloads_run_on_the_cpu_backend():
    """`bench.py --workload env | eval | lookahead` (BASELINE configs 2, 3 and the look-ahead values line) end to end on small batches."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, G2048_BACKEND='cpu', G2048_CPU_THREADS='4')
    for workload, unit in (('env', 'board-steps/s'), ('eval', 'boards/s'), ('lookahead', 'positions/s')):
        res = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--workload', workload, '--batch', '512', '--steps', '8', '--warmup', '8', '--n-tuple', '4'],
                             capture_output=True, text=True, timeout=300, env=env)
        assert res.returncode == 0, res.stderr[-1500:]
        line = json.loads(res.stdout.strip().splitlines()[-1])
        assert line['unit'] == unit and line['value'] > 0 and line['batch'] == 512
        if workload == 'lookahead':
            assert line['finite'] and line['leaf_slots_per_s'] > line['value']
