"""The multi-rank path on CPU: two gloo processes, episodes sharded by lane range, one sum all-reduce of the weight
deltas per epoch (2048_amd/parallel.py).  The table arithmetic on each rank is done by the oracle here (this is a
test of the host-side protocol, which is backend-agnostic; on the GPU box the same DeltaSync drives the Engine)."""
import importlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ref_batch as rb
from tests.golden import formulas

parallel = importlib.import_module('2048_amd.parallel')
N_TUPLE = 2


class HostTable:
    """Stand-in for the device table with the same delta protocol as the Engine."""

    def __init__(self, w):
        self.w = w.astype(np.float64)
        self.w0 = self.w.copy()
        self.slots = len(w)
        self.device = 'cpu'

    def delta_begin(self):
        self.w0 = self.w.copy()

    def delta_extract_into(self, tensor):
        tensor.copy_(torch.from_numpy((self.w - self.w0).astype(np.float32)))

    def delta_apply_from(self, tensor):
        self.w = self.w0 + tensor.numpy().astype(np.float64)
        self.w0 = self.w.copy()


def records(total):
    r = np.random.RandomState(7)
    states = (r.randint(0, 8, (total, 4, 4)) * (r.rand(total, 4, 4) < 0.6)).astype(np.uint8)
    dw = (r.randint(-64, 64, total) * 2.0 ** -10)
    return states, dw


def worker(rank, world, port, total, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    table = HostTable(formulas.weights(N_TUPLE))
    sync = parallel.DeltaSync(table, dist)
    sync.begin()
    states, dw = records(total)
    for epoch in range(2):                                   # two epochs: the snapshot must move with the table
        lane0, count = parallel.shard_lanes(total // 2, rank, world)
        lo = epoch * (total // 2) + lane0
        rb.update(N_TUPLE, table.w, states[lo:lo + count], dw[lo:lo + count])
        sync.all_reduce()
    stats = parallel.reduce_stats(dict(episodes=rank + 1, moves=10 * (rank + 1), score_sum=100, overflow16=0,
                                       best_score=50 * (rank + 1), max_tile=[rank] * 20), dist)
    np.save(os.path.join(out_dir, f'w{rank}.npy'), table.w)
    if rank == 0:
        assert stats['episodes'] == 3 and stats['moves'] == 30 and stats['best_score'] == 100 and stats['max_tile'][0] == 1
        assert sync.reduces == 2
    dist.destroy_process_group()


def test_two_ranks_equal_one_process(tmp_path):
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    total = 2000
    mp.spawn(worker, args=(2, port, total, str(tmp_path)), nprocs=2, join=True)
    w0, w1 = np.load(tmp_path / 'w0.npy'), np.load(tmp_path / 'w1.npy')
    assert np.array_equal(w0, w1)                            # replicas stay identical
    ref = formulas.weights(N_TUPLE).astype(np.float64)
    states, dw = records(total)
    rb.update(N_TUPLE, ref, states, dw)
    assert np.abs(w0 - ref).max() < 1e-5                     # deltas travel as fp32
    assert np.abs(w0 - formulas.weights(N_TUPLE)).max() > 0.1
