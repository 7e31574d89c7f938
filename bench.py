#!/usr/bin/env python3
"""bench.py — board-steps/sec of the batched TD(0) hot path on N MI355X GPUs.

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU.  Under a launcher (torch.distributed.run sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*)
this process IS a rank; without one it starts N rank processes itself before anything touches a GPU and relays
rank 0's line.

Workload (BASELINE.json config 4, the configuration the metric is quoted on): the full TD(0) loop of
QAgent.episode (r_learning.py:228-249) — 4-direction move, n=5 tuple gather + greedy select, TD target, 8-symmetry
scatter-add, spawn, terminal check / auto-reset — on 2^20 concurrent episodes PER GPU (weak scaling: lanes are
sharded by rank, no data-path collective; every --epoch steps the accumulated fp32 weight deltas are sum-all-reduced
with RCCL).  A "step" is one board-step of every lane.

Protocol: fresh games are first advanced --condition steps (untimed, reported, independent of
--warmup) so that the tile distribution is mid-game-like; then W untimed warm-up steps; then --repeats timed regions of
exactly K steps each, every one bracketed by barrier + device synchronisation, time = MAX over ranks; `value` and
`ms_per_step` are the MEDIAN region.  Prints ONE JSON line on rank 0.
"""
import argparse
import importlib
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0                      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
NUM_FEAT = {2: 24, 3: 52, 4: 17, 5: 21, 6: 33}


def algorithmic_bytes(n):
    """Per board-step, SURVEY.md §8(d): env 72 + gathers 4*F*4 + scatter 8*F*(4R+4W) + carry 2*20.
    Split by kernel: k_td_play = env + gathers + carry write; the update kernels = scatter + carry read."""
    F = NUM_FEAT[n]
    play = 72 + 4 * F * 4 + 20
    update = 8 * F * 8 + 20
    return play, update


def cpu_baseline(n, seconds):
    """The oracle's reference-structured scalar port (oracle/ref_scalar.py: dict row table, rot90, NumPy f_n,
    list-of-lists float64 weights, per-move update over 8 symmetries), one core, whole episodes until
    `seconds` of CPU work have passed.  Timed here only; never part of the measured GPU path."""
    from oracle import ref_scalar as rs
    pkg = importlib.import_module('2048_amd')
    rs.row_table()
    np.random.seed(0)
    agent = rs.Agent(n=n, alpha=0.25)
    lane = pkg.rng.LaneRng(2048, 0)

    def draws(n_empty):
        return pkg.rng.spawn_draw(lane.next(), n_empty)
    moves, games = 0, 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        _, _, m = agent.episode(draws)
        moves += m
        games += 1
    dt = time.perf_counter() - t0
    return dict(value=moves / dt, unit='board-steps/s', cores=1, kind='port',
                sample=f'{games} whole TD(0) episodes, n={n}, batch 1, {moves} board-steps in {dt:.1f} s '
                       f'(oracle/ref_scalar.py, NumPy {np.__version__}, host has {os.cpu_count()} logical cores)')


def measured_copy_gbps(device):
    """The box's own device-to-device copy rate (read + write bytes per second) — the practical HBM ceiling next to the
    8 TB/s datasheet peak."""
    try:
        import torch
        dev = torch.device('cuda', device)
        a = torch.empty(1 << 28, dtype=torch.uint8, device=dev)        # 256 MiB, beyond the Infinity Cache with its copy
        b = torch.empty_like(a)
        for _ in range(2):
            b.copy_(a)
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        for _ in range(10):
            b.copy_(a)
        stop.record()
        stop.synchronize()
        return 2 * a.numel() * 10 / (start.elapsed_time(stop) * 1e-3) / 1e9
    except Exception as e:              # never let the side measurement break the benchmark line
        print(f'[bench] copy-bandwidth measurement skipped: {e!r}', file=sys.stderr)
        return None


def side_workload(pkg, args):
    """BASELINE configs 2 and 3 (parity-test cases, measured for DESIGN.md; not the bench metric)."""
    if args.workload == 'env':
        B = 65536 if args.batch == 1 << 20 else args.batch
        eng = pkg.Engine(B, n=0, seed=2048)
        eng.step_random(args.warmup)
        eng.sync()
        eng.timer_start()
        eng.step_random(args.steps)                # one launch, lane state in registers for all `steps`
        ms = eng.timer_stop()
        out = dict(workload='BASELINE config 2: env step only, random valid direction', batch=B, steps=args.steps,
                   value=B * args.steps / (ms * 1e-3), unit='board-steps/s', ms_per_step=ms / args.steps,
                   algorithmic_GBps=72 * B * args.steps / (ms * 1e-3) / 1e9)
    else:
        B = 262144 if args.batch == 1 << 20 else args.batch
        n = 3 if args.n_tuple == 5 else args.n_tuple
        eng = pkg.Engine(B, n=n, seed=2048)
        eng.step_random(64)
        eng.init_weights(seed=7, scale=0.01)
        for _ in range(args.warmup):
            eng.eval_select_device_only()
        eng.sync()
        eng.timer_start()
        for _ in range(args.steps):
            eng.eval_select_device_only()
        ms = eng.timer_stop()
        by = 16 + 4 * NUM_FEAT[n] * 4 + 5
        out = dict(workload=f'BASELINE config 3: {n}-tuple evaluate + 4-way greedy select, inference only', batch=B,
                   steps=args.steps, value=B * args.steps / (ms * 1e-3), unit='boards/s', ms_per_step=ms / args.steps,
                   algorithmic_GBps=by * B * args.steps / (ms * 1e-3) / 1e9)
    emit(out)
    eng.close()


_RESULT_FD = None


def claim_stdout():
    """The contract is ONE line on stdout.  Libraries print there too (RCCL writes its version banner on communicator
    creation), so the process's fd 1 is pointed at stderr for the whole run and the result line goes out through a saved
    duplicate of the original stdout."""
    global _RESULT_FD
    if _RESULT_FD is None:
        sys.stdout.flush()
        _RESULT_FD = os.dup(1)
        os.dup2(2, 1)


def emit(out):
    line = (json.dumps(out) + '\n').encode()
    os.write(_RESULT_FD if _RESULT_FD is not None else 1, line)


def _tail(path, lines=40):
    try:
        with open(path, 'rb') as f:
            return b''.join(f.readlines()[-lines:]).decode(errors='replace')
    except OSError:
        return ''


def self_launch(args):
    """--gpus N without a launcher: start the N rank processes (fresh interpreters, so no GPU state is inherited — this
    parent never imports torch or the HIP library and nothing is re-exec'ed), relay rank 0's stdout.

    The parent watches ALL ranks: the first one that exits non-zero ends the run — the others are terminated (a rank whose
    peer died would otherwise sit in a collective until the backend's own timeout), that rank's stderr tail is printed and
    the parent exits non-zero; --deadline bounds the whole run the same way.  Every rank's stderr goes to a file under a
    temporary directory and is copied to the parent's stderr at the end."""
    import shutil
    import tempfile
    import ctypes
    tmp = tempfile.mkdtemp(prefix='g2048_bench_')
    t_start = time.monotonic()
    libc = ctypes.CDLL(None)

    def die_with_parent():
        libc.prctl(1, 15)                          # PR_SET_PDEATHSIG, SIGTERM

    def launch():
        s = socket.socket()
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
        s.close()
        procs, files = [], []
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                       MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
            err = open(os.path.join(tmp, f'rank{r}.err'), 'wb')
            out = open(os.path.join(tmp, f'rank{r}.out'), 'wb')
            files += [err, out]
            # same process group as the parent (a kill of the group takes the ranks along), and SIGTERM if the parent dies
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=out, stderr=err, preexec_fn=die_with_parent))
        return procs, files

    def stop(procs):
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_end = time.monotonic() + 5.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()

    verdict, attempt = None, 0
    while verdict is None:
        attempt += 1
        procs, files = launch()
        t_launch = time.monotonic()
        while True:
            codes = [p.poll() for p in procs]
            bad = [r for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                stop(procs)
                r = bad[0]
                tail = _tail(os.path.join(tmp, f'rank{r}.err'))
                # the rendezvous port was picked by bind + close: another process may have taken it in between
                if attempt == 1 and time.monotonic() - t_launch < 30 and ('EADDRINUSE' in tail or 'ddress already in use' in tail):
                    print(f'[bench] rendezvous port was taken, launching once more', file=sys.stderr)
                    break
                verdict = f'rank {r} exited with code {codes[r]} (the other ranks were terminated); its stderr tail:\n{tail}'
                break
            if all(c == 0 for c in codes):
                verdict = ''
                break
            if time.monotonic() - t_start > args.deadline:
                stop(procs)
                alive = [r for r, c in enumerate(codes) if c is None]
                verdict = (f'--deadline {args.deadline:.0f} s passed with ranks {alive} still running (terminated); stderr tails:\n'
                           + '\n'.join(f'--- rank {r}\n' + _tail(os.path.join(tmp, f'rank{r}.err'), 15) for r in alive))
                break
            time.sleep(0.05)
        for f in files:
            f.close()
    for r in range(args.gpus):                     # the ranks' diagnostics, in rank order
        sys.stderr.write(_tail(os.path.join(tmp, f'rank{r}.err'), 200 if verdict else 50))
    line = _tail(os.path.join(tmp, 'rank0.out'), 5)
    shutil.rmtree(tmp, ignore_errors=True)
    if verdict:
        raise SystemExit('[bench] FAILED: ' + verdict)
    sys.stdout.write(line)
    sys.stdout.flush()


def fault_inject(spec, rank, where):
    """Test hook (tests/test_cpu_host.py, tests/test_gpu_multi.py): --fault-inject 'exit@1:init,hang@0:init' makes rank 1
    exit with code 3 and rank 0 sleep at the named point, so that the launcher's failure handling can be exercised."""
    for item in filter(None, (spec or '').split(',')):
        what, _, rest = item.partition('@')
        r, _, at = rest.partition(':')
        if int(r) == rank and (at or 'init') == where:
            if what == 'exit':
                print(f'[bench] rank {rank}: injected failure at {where}', file=sys.stderr)
                sys.stderr.flush()
                os._exit(3)
            if what == 'hang':
                time.sleep(3600)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=64)
    ap.add_argument('--repeats', type=int, default=3, help='timed regions of --steps steps each; the median is reported')
    ap.add_argument('--condition', type=int, default=256,
                    help='untimed TD steps that age the fresh boards before --warmup (SURVEY.md 8d input conditioning)')
    ap.add_argument('--batch', type=int, default=1 << 20, help='lanes per GPU')
    ap.add_argument('--n-tuple', type=int, default=5)
    ap.add_argument('--alpha', type=float, default=0.25, help='reference alpha; scaled by the batch rule below')
    ap.add_argument('--epoch', type=int, default=50, help='steps between weight-delta all-reduces (N > 1)')
    ap.add_argument('--cpu-seconds', type=float, default=15.0)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-mean-line', action='store_true', help='skip the secondary measurement of the per-slot mean rule')
    ap.add_argument('--workload', default='td', choices=['td', 'env', 'eval'],
                    help='td = BASELINE config 4 (the metric); env = config 2 (65 536 lanes, env step only); eval = config 3 '
                         '(262 144 lanes, n=3 evaluate + greedy select); env/eval print a reduced JSON line')
    ap.add_argument('--rule', default='sum', choices=['sum', 'mean'],
                    help="how a step's records are applied: 'sum' = the reference's arithmetic (the metric); 'mean' = per-slot "
                         "mean (what QAgent uses for batched training)")
    ap.add_argument('--sync-at-one', action='store_true',
                    help='with one rank, still build the communicator and run the per-epoch delta all-reduce (exercises the RCCL path on a 1-GPU box)')
    ap.add_argument('--comm', default='native', choices=['native', 'torch'],
                    help='delta all-reduce: native = g2048_allreduce_deltas (RCCL on the engine stream; falls back to torch if it '
                         'cannot be set up), torch = torch.distributed all_reduce')
    ap.add_argument('--deadline', type=float, default=900.0,
                    help='self-launched N > 1 runs: seconds after which the parent terminates every rank and fails')
    ap.add_argument('--fault-inject', default='', help=argparse.SUPPRESS)
    ap.add_argument('--backend', default='nccl', help='torch.distributed backend for control traffic (nccl = RCCL; gloo to rehearse on one GPU)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return self_launch(args)

    claim_stdout()
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    n, B, K, W = args.n_tuple, args.batch, args.steps, args.warmup
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')

    dist = None
    if world > 1 or args.sync_at_one:
        import torch
        import torch.distributed as dist
        if world == 1:                                      # --sync-at-one without a launcher
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29517')
            os.environ.setdefault('RANK', '0')
            os.environ.setdefault('WORLD_SIZE', '1')
        import datetime
        limit = datetime.timedelta(seconds=max(60.0, min(args.deadline, 600.0)))    # a lost peer fails the collective, not the day
        fault_inject(args.fault_inject, rank, 'start')
        if args.backend == 'nccl':
            torch.cuda.set_device(local_rank)
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank), timeout=limit)   # RCCL over xGMI
        else:
            if torch.cuda.device_count():
                local_rank = local_rank % torch.cuda.device_count()
                torch.cuda.set_device(local_rank)
            dist.init_process_group(args.backend, timeout=limit)
        fault_inject(args.fault_inject, rank, 'init')

    # torch first: it must initialise its HIP runtime before lib2048_hip.so brings up its own
    copy_gbps = measured_copy_gbps(local_rank) if args.workload == 'td' else None
    pkg = importlib.import_module('2048_amd')
    par = importlib.import_module('2048_amd.parallel')
    if args.workload != 'td':
        return side_workload(pkg, args)
    eng = pkg.Engine(B, n=n, seed=2048, lane0=rank * B, device=local_rank)
    eng.init_weights(seed=7, scale=0.01)                   # same table on every rank (counter-based init)
    # batch rule: every lane adds its delta in the same step and one slot can be hit by all 8 images of every
    # lane, so the reference's per-game alpha is divided by 8 * (concurrent episodes) / num_feat
    alpha = args.alpha * NUM_FEAT[n] / (8.0 * B * world)
    if args.rule == 'mean':
        eng.set_update_rule(1)
        alpha = args.alpha

    sync, comm_kind = None, None
    if dist:
        sync, kind = par.make_sync(eng, dist, rank, world, rule=args.rule, comm=args.comm,
                                   log=lambda m: print(f'[bench] {m}', file=sys.stderr))
        comm_kind = ('native g2048_allreduce_deltas (ncclAllReduce on the engine stream)' if kind == 'native'
                     else f'torch.distributed all_reduce ({args.backend})')
    fault_inject(args.fault_inject, rank, 'comm')

    def run(steps):
        par.run_epochs(eng, sync, alpha, steps, args.epoch)

    def barrier():
        eng.sync()
        if dist:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(x):
        if not dist:
            return x
        import torch
        t = torch.tensor([x], dtype=torch.float64, device='cuda' if args.backend == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    if sync:
        sync.begin()
    run(args.condition)                                    # input conditioning: untimed, not part of --warmup
    run(W)
    fault_inject(args.fault_inject, rank, 'run')
    times, ev_times = [], []
    for _ in range(max(1, args.repeats)):
        barrier()
        t0 = time.perf_counter()
        eng.timer_start()
        run(K)
        ev_ms = eng.timer_stop()
        barrier()
        times.append(max_over_ranks(time.perf_counter() - t0))
        ev_times.append(ev_ms)
    dt = statistics.median(times)
    st = eng.stats()

    # per-kernel launch durations (HIP events on the context's stream), after the timed regions
    ms_play, ms_owner, ms_tail, ms_apply = eng.td_steps_kernel_ms(alpha, 20)
    by_play, by_update = algorithmic_bytes(n)
    kernels = {f'k_td_play<{n}>': ms_play, f'k_td_update_owner<{n}>': ms_owner, 'k_apply_orbits': ms_apply}
    if n == 6:
        kernels['k_td_update_tail<6>'] = ms_tail
    dominant = max(kernels, key=kernels.get)
    traffic_all = {}
    tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
    if os.path.exists(tpath):
        with open(tpath) as f:
            traffic_all = json.load(f)
    # `achieved`: algorithmic bytes of k_td_play's launch (72 + 4 F 4 + 20 per board-step) over its measured
    # duration.  The update kernels remove most of their algorithmic bytes (orbit and coset reductions, LDS accumulation):
    # a fraction is never computed from bytes a kernel does not move.
    ms_step = dt / K * 1e3
    # (k_td_play is quoted even if an update kernel should ever be the longest: it is the only kernel whose algorithmic
    # bytes are bytes it moves; `longest_kernel` says which one took the most time)
    r_kernel, r_bytes, r_ms = f'k_td_play<{n}>', by_play * B, ms_play
    traffic_step = sum(v for k, v in traffic_all.items() if k.endswith(f'_b{B}') and isinstance(v, (int, float))) or None
    achieved = r_bytes / (r_ms * 1e-3) / 1e9
    roofline = {'bound': 'hbm', 'kernel': r_kernel, 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic_all.get(f'{r_kernel}_b{B}'), 'longest_kernel': dominant,
                'algorithmic_bytes_per_launch': r_bytes, 'ms_per_launch': r_ms,
                'ms_kernels': kernels,
                'k_td_play': {'algorithmic_bytes_per_launch': by_play * B, 'ms': ms_play,
                              'GBps': by_play * B / (ms_play * 1e-3) / 1e9, 'frac': by_play * B / (ms_play * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              # its own bound is not HBM: 4 x num_feat divergent 4-byte gathers per lane, one cache line per
                              # cycle through each CU's address path
                              'limited_by': 'L1 misses of the cold table gathers served by L2 (the table is cache-resident; the hot entries come from LDS), VALU issue and per-block latency; DESIGN.md section 4'},
                # the survey's per-step figure (1 792 B at n = 5) over the step time.  NOT a fraction of a bound: most of those
                # bytes never reach HBM — the table and the records are cache-resident, the orbit and coset reductions drop 27
                # of the reference's 48 adds per feature group, a third of the gathers come from LDS — so it can exceed the peak.
                'whole_step': {'algorithmic_bytes': (by_play + by_update) * B, 'ms': ms_step,
                               'GBps': (by_play + by_update) * B / (ms_step * 1e-3) / 1e9,
                               'ratio_to_hbm_peak': (by_play + by_update) * B / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               'hbm_bytes_measured': traffic_step},
                'measured_copy_GBps': copy_gbps}
    if roofline['frac'] > 1.0 or roofline['k_td_play']['frac'] > 1.0:      # cannot happen for bytes a kernel really moves
        roofline['invalid'] = 'fraction above 1: bookkeeping error'

    comm = None
    if sync:
        # the exchange on its own: barrier, one all-reduce of the epoch's delta + its apply pass, device drained; MAX over ranks
        xs = []
        for _ in range(3):
            barrier()
            t0 = time.perf_counter()
            sync.all_reduce()
            eng.sync()
            xs.append(max_over_ranks(time.perf_counter() - t0) * 1e3)
        payload = eng.slots * 4 * (2 if args.rule == 'mean' else 1)
        seen = sync.info() if isinstance(sync, par.NativeSync) else (dist.get_rank(), dist.get_world_size())
        link = 153e9                                       # one xGMI link, MI355X_MICROARCH.md
        comm = {'kind': comm_kind, 'rank_seen': seen[0], 'nranks_seen': seen[1], 'nranks_expected': world,
                'payload_bytes': payload, 'epoch_steps': args.epoch, 'exchanges_per_timed_region': -(-K // args.epoch),
                'allreduce_plus_apply_ms': statistics.median(xs), 'allreduce_plus_apply_ms_repeats': xs,
                'predicted_allreduce_ms': {'ring_one_link': 2 * (world - 1) / world * payload / link * 1e3,
                                           'all_links_direct': (2 * payload / world / link * 1e3) if world > 1 else 0.0}}
        if seen[1] != world:
            raise SystemExit(f'[bench] the communicator reports {seen[1]} ranks, expected {world}')

    mean_line = None
    if world == 1 and not dist and args.rule == 'sum' and not args.no_mean_line:
        # the rule the product trains with (QAgent(batch > 1) -> g2048_set_update_rule(1)), same lanes, alpha unscaled
        eng.set_update_rule(1)
        eng.td_steps(args.alpha, W)
        eng.sync()
        mt = []
        for _ in range(max(1, args.repeats)):
            eng.timer_start()
            eng.td_steps(args.alpha, K)
            mt.append(eng.timer_stop())
        mm = statistics.median(mt)
        mean_line = {'update_rule': 'mean', 'value': B * K / (mm * 1e-3), 'unit': 'board-steps/s', 'ms_per_step': mm / K,
                     'alpha': args.alpha}

    if rank == 0:
        out = {
            'metric': 'board-steps/sec at batch 2^20 (full TD(0) step: move x4, n-tuple gather, greedy select, 8-symmetry scatter-add, spawn)',
            'value': world * B * K / dt,
            'unit': 'board-steps/s',
            'n_gpus': world, 'steps': K, 'warmup': W,
            'ms_per_step': ms_step,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'u8 boards / int32 scores / f32 weights', 'data': 'synthetic',
            'config': {'workload': f'BASELINE config 4: full TD(0) loop, {n}-tuple table ({eng.slots * 4} B), '
                                   f'{B} concurrent episodes per GPU, auto-reset',
                       'batch_per_gpu': B, 'n_tuple': n, 'alpha_effective': alpha, 'update_rule': args.rule,
                       'parallelism': f'episodes sharded over {world} GPU(s)' + (f', weight-delta all-reduce every {args.epoch} steps via {comm_kind}' if sync else '')},
            'protocol': {'conditioning_steps': args.condition, 'repeats': len(times), 'statistic': 'median',
                         'ms_per_step_repeats': [t / K * 1e3 for t in times],
                         'hip_event_ms_per_step_repeats': [t / K for t in ev_times]},
            'roofline': roofline,
            'mean_valid_directions': st['valid_dirs'] / max(1, st['moves']),
            'episodes_finished': st['episodes'],
            'mean_score': st['score_sum'] / max(1, st['episodes']),
        }
        if mean_line:
            out['mean_rule'] = mean_line
        if comm:
            out['comm'] = comm
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(n, args.cpu_seconds)
        emit(out)
    if isinstance(sync, par.NativeSync):
        sync.close()
    eng.close()
    if dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
