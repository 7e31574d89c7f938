"""The multi-rank path on CPU: two gloo processes, episodes sharded by lane range, one all-reduce of the accumulated
weight deltas per epoch, driven by the SAME epoch loop as bench.py and QAgent.train_run (parallel.run_epochs).  The
table arithmetic on each rank is done by the oracle here (a test of the host-side protocol and of the cross-rank
rule, which are transport-agnostic; on the GPU box the same loop drives the Engine: tests/test_gpu_multi.py)."""
import importlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ref_batch as rb
from tests import helpers
from tests.golden import formulas

parallel = importlib.import_module('2048_amd.parallel')
rng_spec = importlib.import_module('2048_amd.rng')
N_TUPLE = 2


class HostStepper:
    """Stand-in for the Engine: a shard of lanes stepped by the float64 oracle (rb.td_step), with the Engine's delta
    protocol (accumulated delta, not W - W0)."""

    def __init__(self, w, lane0, count, rule, seed=2048):
        self.w = w.astype(np.float64)
        self.w0 = self.w.copy()
        self.acc = np.zeros_like(self.w)
        self.slots = len(w)
        self.device = 'cpu'
        self.rule = rule
        self.rng = rng_spec.seed_lanes(seed, lane0, count)
        idx = np.arange(count)
        self.lanes = rb.Lanes(helpers.oracle_new_games(self.rng, idx), np.zeros(count, np.int32))
        self.draws = helpers.SpecDraws(self.rng)
        self.tracking = False

    # stepping
    def td_steps(self, alpha, nsteps):
        for _ in range(nsteps):
            before = self.w.copy()
            out = rb.td_step(N_TUPLE, self.w, self.lanes, alpha, self.draws, self.rule)
            fin = np.nonzero(self.lanes.done)[0]
            if len(fin):                       # auto-reset, as the device does
                self.lanes.boards[fin] = helpers.oracle_new_games(self.rng, fin)
                self.lanes.scores[fin] = 0
                self.lanes.label[fin] = 0.0
                self.lanes.has_prev[fin] = False
                self.lanes.done[fin] = False
            del out
            if self.tracking:
                self.acc += self.w - before    # what this step added (the device mirrors every add into D)

    # delta protocol (DeltaSync's table side)
    def delta_begin(self):
        self.w0 = self.w.copy()
        self.acc[:] = 0
        self.tracking = True

    def delta_extract_into(self, tensor):
        tensor.copy_(torch.from_numpy(self.acc.astype(np.float32)))

    def delta_apply_from(self, tensor):
        self.w = self.w0 + tensor.numpy().astype(np.float64)
        self.w0 = self.w.copy()
        self.acc[:] = 0

    def delta_pack_touched_into(self, tensor2):
        d = self.acc.astype(np.float32)
        tensor2[:self.slots].copy_(torch.from_numpy(d))
        tensor2[self.slots:].copy_(torch.from_numpy((d != 0).astype(np.float32)))

    def delta_apply_mean_from(self, tensor2):
        p = tensor2.numpy().astype(np.float64)
        self.w = self.w0 + p[:self.slots] / np.maximum(1.0, p[self.slots:])
        self.w0 = self.w.copy()
        self.acc[:] = 0


STEPS, EPOCH, LANES = 7, 3, 24          # 3 epochs, the last one shorter


def alpha_of(rule, world):
    return 0.25 if rule == 'mean' else 0.25 * 24 / (64.0 * LANES)        # (small enough for a tame trajectory)


def worker(rank, world, port, rule, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    lane0, count = parallel.shard_lanes(LANES, rank, world)
    table = HostStepper(formulas.weights(N_TUPLE), lane0, count, rule)
    sync = parallel.DeltaSync(table, dist, rule=rule)
    sync.begin()
    done = parallel.run_epochs(table, sync, alpha_of(rule, world), STEPS, EPOCH)
    stats = parallel.reduce_stats(dict(episodes=rank + 1, moves=10 * (rank + 1), score_sum=100, overflow16=0,
                                       best_score=50 * (rank + 1), max_tile=[rank] * 20), dist)
    np.save(os.path.join(out_dir, f'w{rank}.npy'), table.w)
    np.save(os.path.join(out_dir, f'b{rank}.npy'), table.lanes.boards)
    if rank == 0:
        assert done == STEPS and sync.reduces == 3
        assert stats['episodes'] == 3 and stats['moves'] == 30 and stats['best_score'] == 100 and stats['max_tile'][0] == 1
    dist.destroy_process_group()


def expected(rule, world):
    """The same job in ONE process: `world` shards stepped in turn, deltas combined by parallel.combine_deltas."""
    shards = [HostStepper(formulas.weights(N_TUPLE), *parallel.shard_lanes(LANES, r, world), rule) for r in range(world)]
    for s in shards:
        s.delta_begin()
    done = 0
    while done < STEPS:
        chunk = min(EPOCH, STEPS - done)
        for s in shards:
            s.td_steps(alpha_of(rule, world), chunk)
        w = parallel.combine_deltas(shards[0].w0, [s.acc.astype(np.float32) for s in shards], rule, wire=np.float32)
        for s in shards:
            s.w = w.copy()
            s.w0 = w.copy()
            s.acc[:] = 0
        done += chunk
    return shards


@pytest.mark.parametrize('rule', ['sum', 'mean'])
def test_two_ranks_equal_one_process(tmp_path, rule):
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(worker, args=(2, port, rule, str(tmp_path)), nprocs=2, join=True)
    w0, w1 = np.load(tmp_path / 'w0.npy'), np.load(tmp_path / 'w1.npy')
    assert np.array_equal(w0, w1)                            # replicas stay identical
    shards = expected(rule, 2)
    assert np.array_equal(w0, shards[0].w)                   # deltas travel as fp32; two ranks: one rounding, modelled exactly
    for r in range(2):                                       # and the episodes are the same games
        assert np.array_equal(np.load(tmp_path / f'b{r}.npy'), shards[r].lanes.boards)
    assert np.abs(w0 - formulas.weights(N_TUPLE)).max() > 1e-3


def test_mean_rule_across_ranks_is_not_a_sum():
    """A slot moved by both ranks takes the mean of their deltas, a slot moved by one rank its whole delta."""
    w0 = np.zeros(4)
    out = parallel.combine_deltas(w0, [np.array([1.0, 2.0, 0.0, 0.0]), np.array([3.0, 0.0, 5.0, 0.0])], 'mean')
    assert np.array_equal(out, [2.0, 2.0, 5.0, 0.0])
    out = parallel.combine_deltas(w0, [np.array([1.0, 2.0, 0.0, 0.0]), np.array([3.0, 0.0, 5.0, 0.0])], 'sum')
    assert np.array_equal(out, [4.0, 2.0, 5.0, 0.0])


def test_shard_lanes_cover_everything():
    for total in (1, 7, 24, 1 << 20):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_lanes(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == total
            for (a, ca), (b, _) in zip(spans, spans[1:]):
                assert a + ca == b


# ---- NativeSync's set-up protocol: a failure on ONE rank must leave every rank in the same place (rank 0
# raised before the id broadcast the other ranks were already waiting in).  The engine is a stand-in here (no GPU); the
# protocol under test is host code.
class FakeEngine:
    def __init__(self, rank, fail):
        self.rank, self.fail, self.inited, self.destroyed = rank, fail, False, False

    def comm_unique_id(self):
        if self.fail == ('load', self.rank):
            raise OSError('librccl not found (injected)')
        return bytes([self.rank + 1]) * 128

    def comm_init(self, rank, world, uid):
        assert uid == bytes([1]) * 128, 'rank 0\'s id reaches every rank'
        if self.fail == ('init', self.rank):
            raise RuntimeError('ncclCommInitRank refused (injected)')
        self.inited = True

    def comm_destroy(self):
        self.destroyed = True

    def comm_info(self):
        return self.rank, 2


def native_worker(rank, world, port, fail, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import datetime
    dist.init_process_group('gloo', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    eng = FakeEngine(rank, fail)
    raised = False
    try:
        parallel.NativeSync(eng, rank, world, parallel.broadcast_id_torch(dist), parallel.agree_torch(dist))
    except RuntimeError:
        raised = True
    # whatever happened, the ranks are still in step: the next collective pairs up and gives the right answer
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    assert t.item() == 3.0
    with open(os.path.join(out_dir, f'r{rank}.txt'), 'w') as f:
        f.write(f'{int(raised)} {int(eng.inited)} {int(eng.destroyed)}')
    dist.destroy_process_group()


@pytest.mark.parametrize('fail', [None, ('load', 0), ('load', 1), ('init', 0), ('init', 1)])
def test_native_sync_setup_fails_on_all_ranks_or_none(tmp_path, fail):
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(native_worker, args=(2, port, fail, str(tmp_path)), nprocs=2, join=True)
    res = [tuple(int(x) for x in open(tmp_path / f'r{r}.txt').read().split()) for r in range(2)]
    if fail is None:
        assert res == [(0, 1, 0), (0, 1, 0)]
    else:
        assert res[0][0] == 1 and res[1][0] == 1                 # both ranks raised
        if fail[0] == 'load':
            assert not res[0][1] and not res[1][1]               # nobody entered ncclCommInitRank
        else:
            ok_rank = 1 - fail[1]
            assert res[ok_rank] == (1, 1, 1)                     # the rank that got a communicator gave it back
