"""Multi-GPU data parallelism for the TD(0) learner: one process per GPU, episodes sharded by lane range, the
fp32 weight table replicated, and ONE collective per epoch — a sum all-reduce of the weight deltas over
RCCL/xGMI (torch.distributed backend "nccl"; "gloo" in the CPU tests).  The reference has no counterpart
(single Python thread, application.py:611); SURVEY.md §8(e) defines the scheme.

    sync = DeltaSync(table, dist); sync.begin()
    loop: table.td_steps(alpha, E); sync.all_reduce()

`table` is anything with delta_begin() / delta_extract_into(tensor) / delta_apply_from(tensor) / slots / device:
the Engine on a GPU, or a host stand-in in the gloo tests.
"""
import numpy as np


def shard_lanes(total_lanes, rank, world):
    """Contiguous lane range of `rank`: (lane0, count).  Lanes are seeded by global lane id, so any split of
    the same total gives the same set of episodes."""
    base, extra = divmod(int(total_lanes), int(world))
    count = base + (1 if rank < extra else 0)
    lane0 = rank * base + min(rank, extra)
    return lane0, count


class EngineTable:
    """Adapter: Engine -> the table protocol DeltaSync needs (device pointers from torch tensors)."""

    def __init__(self, engine):
        self.e = engine
        self.slots = engine.slots
        self.device = f'cuda:{engine.device}'

    def delta_begin(self):
        self.e.delta_begin()

    def delta_extract_into(self, tensor):
        self.e.delta_extract(tensor.data_ptr())

    def delta_apply_from(self, tensor):
        self.e.delta_apply(tensor.data_ptr())


class DeltaSync:
    def __init__(self, table, dist, group=None):
        import torch
        self.torch = torch
        self.table = table if hasattr(table, 'delta_extract_into') else EngineTable(table)
        self.dist, self.group = dist, group
        self.buf = torch.zeros(self.table.slots, dtype=torch.float32, device=self.table.device)
        self.reduces = 0

    def begin(self):
        """Snapshot W0 = W: deltas are measured from here."""
        self.table.delta_begin()

    def all_reduce(self):
        """D = W - W0 on every rank; D <- sum over ranks; W = W0 + D; W0 = W (the next epoch starts here)."""
        self.table.delta_extract_into(self.buf)            # synchronises the engine's stream
        if self.buf.is_cuda and self.dist.get_backend(self.group) == 'gloo':
            host = self.buf.cpu()                          # rehearsal on a box without RCCL peers: stage through the host
            self.dist.all_reduce(host, op=self.dist.ReduceOp.SUM, group=self.group)
            self.buf.copy_(host)
        else:
            self.dist.all_reduce(self.buf, op=self.dist.ReduceOp.SUM, group=self.group)
        if self.buf.is_cuda:
            self.torch.cuda.synchronize(self.buf.device)
        self.table.delta_apply_from(self.buf)
        self.reduces += 1


def reduce_stats(stats, dist, device='cpu', group=None):
    """Sum the episode counters of all ranks (best_score: max).  `stats` is Engine.stats()."""
    import torch
    keys = ['episodes', 'moves', 'score_sum', 'overflow16']
    vec = torch.tensor([stats[k] for k in keys] + list(stats['max_tile']), dtype=torch.int64, device=device)
    best = torch.tensor([stats['best_score']], dtype=torch.int64, device=device)
    dist.all_reduce(vec, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(best, op=dist.ReduceOp.MAX, group=group)
    vec = vec.cpu().numpy()
    out = {k: int(v) for k, v in zip(keys, vec[:len(keys)])}
    out['max_tile'] = [int(v) for v in vec[len(keys):]]
    out['best_score'] = int(best.item())
    return out
