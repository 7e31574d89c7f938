#!/usr/bin/env python3
"""bench.py — board-steps/sec of the batched TD(0) hot path on N MI355X GPUs.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

Workload (BASELINE.json config 4, the configuration the metric is quoted on): the full TD(0) loop of
QAgent.episode — 4-direction move, n=5 tuple gather + greedy select, TD target, 8-symmetry scatter-add, spawn,
terminal check / auto-reset — on 2^20 concurrent episodes PER GPU (weak scaling: lanes are sharded by rank, no
data-path collective; every --epoch steps the fp32 weight deltas are sum-all-reduced with RCCL).
A "step" is one board-step of every lane.  Prints ONE JSON line on rank 0.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0                      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
NUM_FEAT = {2: 24, 3: 52, 4: 17, 5: 21, 6: 33}


def algorithmic_bytes(n):
    """Per board-step, SURVEY.md §8(d): env 72 + gathers 4*F*4 + scatter 8*F*(4R+4W) + carry 2*20.
    Split by kernel: k_td_play = env + gathers + carry write; k_td_update = scatter + carry read."""
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
    print(json.dumps(out), flush=True)
    eng.close()


def gather_floor_ms(n, B, device):
    """Time for the table gathers of one k_td_play launch at one lane per cycle per CU (the measured rate of divergent
    4-byte loads through the texture addresser)."""
    import torch
    p = torch.cuda.get_device_properties(device)
    clock_hz = getattr(p, 'clock_rate', 2400000) * 1e3
    return 4.0 * NUM_FEAT[n] * B / (p.multi_processor_count * clock_hz) * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=64)
    ap.add_argument('--batch', type=int, default=1 << 20, help='lanes per GPU')
    ap.add_argument('--n-tuple', type=int, default=5)
    ap.add_argument('--alpha', type=float, default=0.25, help='reference alpha; scaled by the batch rule below')
    ap.add_argument('--epoch', type=int, default=50, help='steps between weight-delta all-reduces (N > 1)')
    ap.add_argument('--cpu-seconds', type=float, default=15.0)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--workload', default='td', choices=['td', 'env', 'eval'],
                    help='td = BASELINE config 4 (the metric); env = config 2 (65 536 lanes, env step only); eval = config 3 '
                         '(262 144 lanes, n=3 evaluate + greedy select); env/eval print a reduced JSON line')
    ap.add_argument('--rule', default='sum', choices=['sum', 'mean'],
                    help="how a step's records are applied: 'sum' = the reference's arithmetic (the metric); 'mean' = per-slot "
                         "mean (what QAgent uses for batched training; a second accumulation pass)")
    ap.add_argument('--sync-at-one', action='store_true',
                    help='with one rank, still create the process group and run the per-epoch delta all-reduce (exercises the RCCL path on a 1-GPU box)')
    ap.add_argument('--backend', default='nccl', help='torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse on one GPU)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    n, B, K, W = args.n_tuple, args.batch, args.steps, args.warmup
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}')

    dist = None
    if world > 1 or args.sync_at_one:
        import torch
        import torch.distributed as dist
        if world == 1:                                      # --sync-at-one without a launcher
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29517')
            os.environ.setdefault('RANK', '0')
            os.environ.setdefault('WORLD_SIZE', '1')
        if args.backend == 'nccl':
            torch.cuda.set_device(local_rank)
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))   # RCCL over xGMI
        else:
            local_rank = local_rank % torch.cuda.device_count()
            torch.cuda.set_device(local_rank)
            dist.init_process_group(args.backend)

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
    sync = par.DeltaSync(eng, dist) if dist else None

    def run(steps):
        done = 0
        while done < steps:
            chunk = min(args.epoch, steps - done) if sync else steps - done
            eng.td_steps(alpha, chunk)
            done += chunk
            if sync:
                sync.all_reduce()

    def barrier():
        eng.sync()
        if dist:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    if sync:
        sync.begin()
    run(W)
    barrier()
    t0 = time.perf_counter()
    eng.timer_start()
    run(K)
    ev_ms = eng.timer_stop()
    barrier()
    dt = time.perf_counter() - t0
    if dist:
        import torch
        t = torch.tensor([dt], dtype=torch.float64, device='cuda' if args.backend == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # per-kernel launch durations (HIP events on the context's stream), after the timed region
    ms_play, ms_update = eng.td_steps_profiled(alpha, 20)
    st = eng.stats()
    by_play, by_update = algorithmic_bytes(n)
    # the update is k_td_update_owner (+ k_apply_orbits, ~1 % of it; + k_td_update_tail for n = 6), timed together
    dominant = 'k_td_update' if ms_update >= ms_play else 'k_td_play'
    dom_ms = max(ms_update, ms_play)
    dom_bytes = (by_update if dominant == 'k_td_update' else by_play) * B
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
    if os.path.exists(tpath):
        with open(tpath) as f:
            traffic = json.load(f).get(f'{dominant}_n{n}_b{B}')

    if rank == 0:
        out = {
            'metric': 'board-steps/sec at batch 2^20 (full TD(0) step: move x4, n-tuple gather, greedy select, 8-symmetry scatter-add, spawn)',
            'value': world * B * K / dt,
            'unit': 'board-steps/s',
            'n_gpus': world, 'steps': K, 'warmup': W,
            'ms_per_step': dt / K * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'u8 boards / int32 scores / f32 weights', 'data': 'synthetic',
            'config': {'workload': f'BASELINE config 4: full TD(0) loop, {n}-tuple table ({eng.slots * 4} B), '
                                   f'{B} concurrent episodes per GPU, auto-reset',
                       'batch_per_gpu': B, 'n_tuple': n, 'alpha_effective': alpha, 'update_rule': args.rule,
                       'parallelism': f'episodes sharded over {world} GPU(s)' + (f', weight-delta all-reduce every {args.epoch} steps' if sync else '')},
            'roofline': {'bound': 'hbm', 'kernel': 'k_td_update_owner' if dominant == 'k_td_update' else dominant, 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                         'algorithmic_bytes_per_launch': dom_bytes, 'ms_per_launch': dom_ms,
                         'ms_k_td_play': ms_play, 'ms_k_td_update': ms_update,
                         'whole_step_algorithmic_GBps': (by_play + by_update) * B / (dt / K) / 1e9,
                         # k_td_play's own bound is not HBM: 4 x num_feat divergent 4-byte gathers per lane pass the CU's address
                         # unit at ~1 lane per cycle; this is that floor for one launch
                         'ms_k_td_play_gather_floor': gather_floor_ms(n, B, local_rank),
                         'measured_copy_GBps': copy_gbps},
            'hip_event_ms_per_step': ev_ms / K,
            'episodes_finished': st['episodes'],
            'mean_score': st['score_sum'] / max(1, st['episodes']),
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(n, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    eng.close()
    if dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
