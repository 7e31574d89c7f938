"""Multi-GPU entry points of the C ABI on ONE GPU (world size 1 through RCCL, two shards exchanged by hand, and
bench.py's own two-rank launch with gloo as the transport): the epoch arithmetic of include/g2048.h "multi-GPU" —
the thing the reference lacks (one Python thread, r_learning.py:269-296) — and a-14 init_weights."""
import importlib
import json
import os
import subprocess
import time
import sys

import numpy as np
import pytest
import torch  # noqa: F401  (before lib2048_hip.so: PyTorch brings its own HIP runtime, which must come up first in a process that uses both)

from tests import helpers
from tests.golden import formulas

pytestmark = pytest.mark.gpu
pkg = importlib.import_module('2048_amd')
importlib.import_module('2048_amd.engine')
parallel = importlib.import_module('2048_amd.parallel')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('n', [2, 5])
def test_weights_init_matches_spec(n):
    """a-14: k_weights_init against its host statement (2048_amd/rng.py) bit for bit, and init_weights' range and
    moments (r_learning.py:139-149: np.random.random(...)/100 -> U[0, 0.01))."""
    eng = pkg.Engine(64, n=n, seed=1)
    eng.init_weights(seed=7, scale=0.01)
    w = eng.get_weights()
    assert np.array_equal(w, pkg.rng.init_weights_np(eng.slots, 7, 0.01))
    assert w.min() >= 0.0 and w.max() < 0.01
    assert abs(w.mean() - 0.005) < 2e-5 and abs(w.std() - 0.01 / np.sqrt(12)) < 2e-5
    offs, sizes = pkg.engine.feature_layout(n)              # no feature's table is biased: action 0 gets no head start (:124-126)
    for o, s in zip(offs, sizes):                           # within 5 standard errors of the mean of U[0, 0.01)
        assert abs(w[o:o + s].mean() - 0.005) < 5 * 0.01 / np.sqrt(12.0 * s)
    eng.init_weights(seed=8, scale=0.01)
    assert not np.array_equal(eng.get_weights(), w)
    eng.close()


@pytest.mark.parametrize('n,rule,algo', [(4, 0, ''), (5, 1, ''), (3, 0, ''), (2, 1, ''), (6, 0, ''), (5, 0, 'rsag'), (4, 1, 'rsag'), (3, 0, 'rsag')])
def test_accumulated_delta_and_native_allreduce_world1(n, rule, algo, monkeypatch):
    """With one rank the epoch exchange must be the identity up to one fp32 rounding per slot: an engine that tracks its
    delta steps exactly like a plain engine (the accumulator only mirrors the adds), the accumulated delta is W - W0, and
    g2048_allreduce_deltas through RCCL (g2048_comm_init with nranks = 1) leaves W = W0 + D.
    algo 'rsag' (G2048_COMM_ALGO, round 4): the same exchange as ncclReduceScatter + ncclAllGather over padded chunks (n = 3's
    212 992 and n = 5's 5 308 416 slots are not multiples of the 256-float chunk granule times every rank count)."""
    import torch
    if algo:
        monkeypatch.setenv('G2048_COMM_ALGO', algo)             # (read at g2048_create)
    B, E = 8192, 6
    alpha = 0.25 if rule else 0.25 * pkg.engine.NUM_FEAT[n] / (8.0 * B)
    w_init = formulas.weights(n, scale=2.0 ** -4)
    plain = pkg.Engine(B, n=n, seed=5)
    plain.set_weights(w_init)
    plain.set_update_rule(rule)
    plain.td_steps(alpha, E)
    ref = plain.get_weights()
    boards_ref = plain.get_boards()
    plain.close()

    eng = pkg.Engine(B, n=n, seed=5)
    eng.set_weights(w_init)
    eng.set_update_rule(rule)
    eng.comm_init(0, 1, pkg.Engine.comm_unique_id())
    eng.delta_begin()
    d = torch.zeros(eng.slots, dtype=torch.float32, device='cuda:0')
    w0 = w_init.astype(np.float64)
    for epoch in range(2):
        eng.td_steps(alpha, E)
        eng.delta_extract(d.data_ptr())
        w_mid = eng.get_weights()
        if epoch == 0:
            if n <= 4:                  # every sum in 64-bit fixed point: repeatable bit for bit
                assert np.array_equal(w_mid, ref), 'tracking the delta must not change the steps'
            else:                       # (all sums are 64-bit fixed point in LDS since round 2; what varies is the order in which the workgroups that share a chunk flush into D with fp32 atomics, and the f_6 bins of n = 6)
                assert np.abs(w_mid.astype(np.float64) - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())
            assert np.array_equal(eng.get_boards(), boards_ref)
        acc = d.cpu().numpy().astype(np.float64)
        assert np.abs(acc).max() > 0
        tol = 4e-7 * np.maximum(1.0, np.maximum(np.abs(w_mid), np.abs(acc))) * (E + 1)
        assert (np.abs(w0 + acc - w_mid) <= tol).all()      # D is what the steps added
        eng.allreduce_deltas()                               # W = W0 + D on the engine's stream
        w0 = eng.get_weights().astype(np.float64)
        assert (np.abs(w0 - w_mid) <= tol).all()
        eng.delta_extract(d.data_ptr())
        assert float(d.abs().max()) == 0.0                   # the next epoch starts from a clean accumulator
    assert eng.allreduce_f64([1.5, 2.0])[1] == 2.0 and eng.allreduce_f64([3.0], op='max')[0] == 3.0
    eng.comm_destroy()
    eng.close()


@pytest.mark.parametrize('rule', ['sum', 'mean'])
def test_two_shards_exchange_by_hand(rule):
    """Two engines = two ranks' lane shards on one GPU; the epoch exchange done with the C ABI's host-driven calls and
    a NumPy sum in place of the all-reduce.  Both replicas must hold parallel.combine_deltas of their deltas."""
    import torch
    n, B, E = 4, 4096, 5
    alpha = 0.25 if rule == 'mean' else 0.25 * 17 / (8.0 * 2 * B)
    w_init = formulas.weights(n, scale=2.0 ** -4)
    engs = []
    for r in range(2):
        e = pkg.Engine(B, n=n, seed=9, lane0=r * B)
        e.set_weights(w_init)
        e.set_update_rule(1 if rule == 'mean' else 0)
        e.delta_begin()
        engs.append(e)
    assert not np.array_equal(engs[0].get_boards(), engs[1].get_boards())     # different episodes
    w0 = w_init.astype(np.float64)
    for epoch in range(2):
        for e in engs:
            e.td_steps(alpha, E)
        bufs = [torch.zeros(e.slots * 2, dtype=torch.float32, device='cuda:0') for e in engs]
        for e, b in zip(engs, bufs):
            e.delta_pack_touched(b.data_ptr())
        deltas = [b[:engs[0].slots].cpu().numpy() for b in bufs]
        touched = [b[engs[0].slots:].cpu().numpy() for b in bufs]
        for d, t in zip(deltas, touched):
            assert np.array_equal(t, (d != 0).astype(np.float32))
        total = (bufs[0] + bufs[1])                          # the all-reduce
        want = parallel.combine_deltas(w0, deltas, rule, wire=np.float32)
        for e in engs:
            if rule == 'mean':
                e.delta_apply_mean(total.data_ptr())
            else:
                e.delta_apply(total[:e.slots].contiguous().data_ptr())
        got = [e.get_weights() for e in engs]
        assert np.array_equal(got[0], got[1])                # replicas identical
        assert np.abs(got[0].astype(np.float64) - want).max() <= 2e-7 * max(1.0, np.abs(want).max())
        w0 = got[0].astype(np.float64)
    assert np.abs(w0 - w_init).max() > 1e-4
    for e in engs:
        e.close()


def test_bench_self_launch_two_ranks_gloo():
    """bench.py --gpus 2 with no launcher: the parent starts two rank processes (both on this box's one GPU, gloo as the
    control and delta transport) that run the same epoch loop the 8-GPU job runs, and relays rank 0's JSON line."""
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--comm', 'torch', '--batch', '65536',
           '--steps', '12', '--warmup', '3', '--condition', '8', '--epoch', '5', '--repeats', '2', '--no-cpu-baseline']
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads(res.stdout.strip().splitlines()[-1])
    assert line['n_gpus'] == 2 and line['steps'] == 12 and line['warmup'] == 3 and line['scaling'] == 'weak'
    assert line['value'] > 0 and 'all-reduce every 5 steps' in line['config']['parallelism']
    assert line['roofline']['frac'] <= 1.0
    assert line['comm']['nranks_seen'] == 2 and line['comm']['allreduce_plus_apply_ms'] > 0


def _bench_two_ranks(*extra, timeout=900):
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--no-cpu-baseline', *extra]
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    t0 = time.monotonic()
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
    return res, time.monotonic() - t0


@pytest.mark.parametrize('rule', ['sum', 'mean'])
def test_bench_two_ranks_n6_exchange(rule):
    """BASELINE config 5's exchange rehearsed with two ranks: the n = 6 table (382.65 MB of deltas per epoch, 765 MB under
    the mean rule) goes through run_epochs' all-reduce.  --comm native: with both ranks on one GPU ncclCommInitRank
    refuses, and BOTH ranks must then fall back to torch.distributed together (parallel.make_sync)."""
    res, _ = _bench_two_ranks('--n-tuple', '6', '--batch', '65536', '--steps', '6', '--warmup', '2', '--condition', '4', '--epoch', '3',
                              '--repeats', '1', '--rule', rule, '--comm', 'native')
    assert res.returncode == 0, res.stderr[-3000:]
    line = json.loads(res.stdout.strip().splitlines()[-1])
    assert line['n_gpus'] == 2 and line['config']['n_tuple'] == 6 and line['value'] > 0
    comm = line['comm']
    assert comm['nranks_seen'] == 2 and comm['exchanges_per_timed_region'] == 2
    assert comm['payload_bytes'] == 95662848 * 4 * (2 if rule == 'mean' else 1)
    assert comm['allreduce_plus_apply_ms'] > 0


def test_bench_rank_killed_mid_run_ends_the_job():
    """One rank dies after the warm-up, between two exchanges: the survivor would wait in the next all-reduce; the parent
    must notice, terminate it and exit non-zero within seconds, without a result line."""
    res, dt = _bench_two_ranks('--batch', '65536', '--steps', '12', '--warmup', '3', '--condition', '8', '--epoch', '5', '--comm', 'torch',
                               '--fault-inject', 'exit@1:run', timeout=300)
    assert res.returncode != 0 and dt < 120
    assert 'rank 1 exited with code 3' in res.stderr and res.stdout.strip() == ''


def test_qagent_train_run_two_ranks(tmp_path):
    """QAgent.train_run as a two-rank job (both ranks on this box's one GPU, gloo carrying the control traffic and the
    deltas): the ranks play different shards, exchange their accumulated deltas every epoch and stay identical replicas
    with identical schedules; rank 0 reports for the job."""
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    dump = str(tmp_path / 'rank')
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'tools', 'train_multi.py'), '--n', '4', '--batch', '8192', '--episodes', '30000',
                                       '--epoch', '32', '--backend', 'gloo', '--comm', 'torch', '--dump', dump], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs[0][1][-1500:] + outs[1][1][-1500:]
    a, b = (np.load(f'{dump}.{r}.npz') for r in range(2))
    assert int(a['step']) == int(b['step']) >= 30000 and float(a['alpha']) == float(b['alpha']) and int(a['top_tile']) == int(b['top_tile'])
    assert np.array_equal(a['history'], b['history']) and len(a['history']) > 5
    assert np.array_equal(a['w_head'], b['w_head']) and np.array_equal(a['w_tail'], b['w_tail']) and float(a['wsum']) == float(b['wsum'])
    assert int(a['reduces']) > 3 and str(a['sync']) == 'DeltaSync'
    assert 'training session started' in outs[0][0] and 'on each of 2 GPUs' in outs[0][0] and 'training session started' not in outs[1][0]
    assert a['history'][-1] > a['history'][0]                 # and it learns
