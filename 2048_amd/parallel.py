"""Multi-GPU data parallelism for the TD(0) learner: one process per GPU, episodes sharded by lane range, the
fp32 weight table replicated, and ONE collective per epoch — a sum all-reduce of the accumulated weight deltas
over RCCL/xGMI.  The reference has no counterpart (single Python thread, r_learning.py:269-296,
application.py:611); SURVEY.md §8(e) defines the scheme, include/g2048.h ("multi-GPU") the arithmetic:

    rule 'sum'  : W = W0 + sum_r D_r
    rule 'mean' : W = W0 + sum_r D_r / max(1, #{r : D_r != 0})      (per slot: the mean over the ranks that moved it)

Two transports, same arithmetic:
    NativeSync  — g2048_comm_init / g2048_allreduce_deltas: ncclAllReduce on the context's own stream, no host wait;
    DeltaSync   — host-driven through torch.distributed (backend "nccl" = RCCL, or "gloo" in the CPU tests).

    sync.begin()
    run_epochs(stepper, sync, alpha, steps, epoch)        # the loop bench.py, QAgent.train_run and the tests share

`stepper` is anything with td_steps(alpha, nsteps): the Engine on a GPU, or a host stand-in in the gloo tests.  A
table for DeltaSync is anything with delta_begin() / delta_extract_into(t) / delta_apply_from(t) (and, for the mean
rule, delta_pack_touched_into(t2) / delta_apply_mean_from(t2)) / slots / device.
"""
import numpy as np


def shard_lanes(total_lanes, rank, world):
    """Contiguous lane range of `rank`: (lane0, count).  Lanes are seeded by global lane id, so any split of
    the same total gives the same set of episodes."""
    base, extra = divmod(int(total_lanes), int(world))
    count = base + (1 if rank < extra else 0)
    lane0 = rank * base + min(rank, extra)
    return lane0, count


def run_epochs(stepper, sync, alpha, steps, epoch):
    """`steps` board-steps of every lane, the weight deltas exchanged every `epoch` steps (and after the last,
    shorter epoch).  sync = None: no exchange (one rank)."""
    done = 0
    while done < steps:
        chunk = min(int(epoch), steps - done) if sync is not None else steps - done
        stepper.td_steps(alpha, chunk)
        done += chunk
        if sync is not None:
            sync.all_reduce()
    return done


class EngineTable:
    """Adapter: Engine -> the table protocol DeltaSync needs (device pointers from torch tensors)."""

    def __init__(self, engine):
        self.e = engine
        self.slots = engine.slots
        self.device = 'cpu' if getattr(engine, 'backend', 'hip') == 'cpu' else f'cuda:{engine.device}'

    def delta_begin(self):
        self.e.delta_begin()

    def delta_extract_into(self, tensor):
        self.e.delta_extract(tensor.data_ptr())

    def delta_apply_from(self, tensor):
        self.e.delta_apply(tensor.data_ptr())

    def delta_pack_touched_into(self, tensor2):
        self.e.delta_pack_touched(tensor2.data_ptr())

    def delta_apply_mean_from(self, tensor2):
        self.e.delta_apply_mean(tensor2.data_ptr())


class DeltaSync:
    """Host-driven epoch exchange through torch.distributed."""

    def __init__(self, table, dist, group=None, rule='sum'):
        import torch
        assert rule in ('sum', 'mean')
        self.torch = torch
        self.table = table if hasattr(table, 'delta_extract_into') else EngineTable(table)
        self.dist, self.group, self.rule = dist, group, rule
        n = self.table.slots * (2 if rule == 'mean' else 1)
        self.buf = torch.zeros(n, dtype=torch.float32, device=self.table.device)
        self.reduces = 0

    def begin(self):
        """Start of the first epoch: W0 = W, accumulated delta = 0."""
        self.table.delta_begin()

    def all_reduce(self):
        """End of an epoch (see the module docstring for the arithmetic); the next epoch starts from the result."""
        if self.rule == 'mean':
            self.table.delta_pack_touched_into(self.buf)   # synchronises the engine's stream
        else:
            self.table.delta_extract_into(self.buf)
        if self.buf.is_cuda and self.dist.get_backend(self.group) == 'gloo':
            host = self.buf.cpu()                          # rehearsal on a box without RCCL peers: stage through the host
            self.dist.all_reduce(host, op=self.dist.ReduceOp.SUM, group=self.group)
            self.buf.copy_(host)
        else:
            self.dist.all_reduce(self.buf, op=self.dist.ReduceOp.SUM, group=self.group)
        if self.buf.is_cuda:
            self.torch.cuda.synchronize(self.buf.device)
        if self.rule == 'mean':
            self.table.delta_apply_mean_from(self.buf)
        else:
            self.table.delta_apply_from(self.buf)
        self.reduces += 1


class NativeSync:
    """The C ABI's own RCCL path: g2048_comm_init once (collective), then g2048_allreduce_deltas per epoch, queued on
    the engine's stream.

    Setting it up is itself a little protocol, and every rank walks through ALL of its collective steps whatever
    happens locally, so that a failure on one rank (librccl missing, ncclCommInitRank refusing) can never leave the
    ranks in different collectives (round 2's version raised on rank 0 before the id broadcast the other ranks
    were already waiting in):
      1. every rank probes the library (g2048_comm_unique_id: loads librccl, creates an id; only rank 0's is used);
      2. `exchange_id(id_or_None) -> id` carries rank 0's 128-byte id (or None) to every rank — any out-of-band
         channel: torch.distributed broadcast, a TCPStore, a file;
      3. `agree(ok) -> bool` (logical AND over the ranks): does everyone have a library and an id?  If not, ALL raise;
      4. g2048_comm_init (ncclCommInitRank, collective), then agree() again: if any rank failed, those that succeeded
         destroy their communicator and ALL raise.
    A caller that catches the exception therefore falls back on every rank or on none."""

    def __init__(self, engine, rank, world, exchange_id, agree=None):
        agree = agree or (lambda ok: ok)
        err = None
        try:
            uid = engine.comm_unique_id()
        except Exception as e:                 # no librccl here: still take part in steps 2 and 3
            uid, err = None, e
        uid = exchange_id(uid if rank == 0 else None)
        if not agree(err is None and uid is not None):
            raise RuntimeError(f'native RCCL path unavailable on at least one rank (rank {rank}: {err!r})')
        try:
            engine.comm_init(rank, world, uid)
        except Exception as e:
            err = e
        if not agree(err is None):
            if err is None:
                engine.comm_destroy()
            raise RuntimeError(f'g2048_comm_init failed on at least one rank (rank {rank}: {err!r})')
        self.e = engine
        self.reduces = 0

    def begin(self):
        self.e.delta_begin()

    def all_reduce(self):
        self.e.allreduce_deltas()                          # the update rule is the context's (g2048_set_update_rule)
        self.reduces += 1

    def info(self):
        """(rank, nranks) as the communicator reports them (ncclCommUserRank / ncclCommCount)."""
        return self.e.comm_info()

    def close(self):
        self.e.comm_destroy()


def broadcast_id_torch(dist, group=None):
    """exchange_id for NativeSync over an initialised torch.distributed group (any backend)."""
    def exchange(uid):
        box = [uid]
        dist.broadcast_object_list(box, src=0, group=group)
        return box[0]
    return exchange


def agree_torch(dist, group=None):
    """agree for NativeSync: logical AND of a host flag over the ranks (object collective: works on any backend
    without a device tensor)."""
    def agree(ok):
        flags = [None] * dist.get_world_size(group)
        dist.all_gather_object(flags, bool(ok), group=group)
        return all(flags)
    return agree


def make_sync(engine, dist, rank, world, rule='sum', comm='native', log=None, group=None):
    """The per-epoch exchange for one rank of a torch.distributed job: the native RCCL path if every rank can set it
    up, otherwise (all ranks together) torch.distributed's all_reduce.  Returns (sync, kind)."""
    if comm == 'native':
        try:
            return NativeSync(engine, rank, world, broadcast_id_torch(dist, group), agree_torch(dist, group)), 'native'
        except Exception as e:
            if log:
                log(f'rank {rank}: {e}; using torch.distributed all_reduce')
    return DeltaSync(engine, dist, group=group, rule=rule), 'torch'


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


def combine_deltas(w0, deltas, rule='sum', wire=None):
    """The epoch arithmetic on host arrays (float64): what every replica holds after the exchange.  Used by the tests
    as the statement of the cross-rank rule.  wire=np.float32 rounds the summed delta as the fp32 all-reduce does
    (exact for two ranks: the float64 sum of two fp32 numbers is exact, so one rounding remains)."""
    d = np.stack([np.asarray(x, np.float64) for x in deltas])
    total = d.sum(axis=0)
    if wire is not None:
        total = total.astype(wire).astype(np.float64)
    if rule == 'sum':
        return np.asarray(w0, np.float64) + total
    touched = (d != 0).sum(axis=0)
    return np.asarray(w0, np.float64) + total / np.maximum(1, touched)
