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
L2_PEAK_GBS = 34500.0                      # MI355X_MICROARCH.md, L2 (per XCD, aggregate): ~34.5 TB/s
LDS_ATOMIC_PEAK = 4.1 * 256 * 2.4e9        # ds_add_u64 lane-adds/s, chip-wide: 4.1 per cycle per CU (profiles/r01_lds_atomic_microbench.txt) x 256 CUs x 2.4 GHz
NUM_FEAT = {2: 24, 3: 52, 4: 17, 5: 21, 6: 33}
ORBIT_ADDS = {2: 24, 3: 52, 4: 17, 5: 21, 6: 21}          # LDS adds per record after the orbit / coset reduction (n = 6: + 16 binned pairs)


def algorithmic_bytes(n):
    """Per board-step, SURVEY.md 8(d): env 72 + gathers 4*F*4 + scatter 8*F*(4R+4W) + carry 2*20 — the REFERENCE algorithm's
    bytes.  Split: k_td_play = env + gathers + carry write.  The scatter term (8 F read-modify-writes of the table) is what the
    reference does, not what this implementation moves (see implementation_bytes); it is quoted for the record only."""
    F = NUM_FEAT[n]
    play = 72 + 4 * F * 4 + 20
    update = 8 * F * 8 + 20
    return play, update


def implementation_bytes(n):
    """Bytes per lane and step that THIS implementation must move through HBM when nothing but the table is cache-resident
     — the denominator of the whole-step fraction:
      k_td_play   reads  board 16 + rng 16 + score 4 + label 4 + flags 1 + lane id 4                    = 45
                  writes board 16 + rng 16 + score 4 + label 4 + flags 1 + last move 2 + afterstate 16
                         + dw 4 + orbit-index record 50 (n >= 4)                                        = 63 (+ 50)
      update      reads  the record once: orbit indices 50 + dw 4 (n >= 4) or afterstate 16 + dw 4      = 54 | 20
      apply       the orbit tables once (read, clear the other buffer): a per-launch constant, quoted apart
    The table gathers and the LDS adds are NOT here: the table is L2 / Infinity-Cache resident by design."""
    rec = 50 if n >= 4 else 0
    play = 45 + 63 + rec
    update = (50 + 4) if n >= 4 else (16 + 4)
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
    out = dict(value=moves / dt, unit='board-steps/s', cores=1, kind='port',
               sample=f'{games} whole TD(0) episodes, n={n}, batch 1, {moves} board-steps in {dt:.1f} s '
                      f'(oracle/ref_scalar.py, NumPy {np.__version__}, host has {os.cpu_count()} logical cores); '
                      f'calibration against the imported reference on one core: BASELINE.md section 4 (port / reference = 0.90 - 1.19)')
    try:
        out['cpu_ref'] = cpu_ref_lines(pkg, n)
    except Exception as e:              # the second CPU line must not take the benchmark line down
        out['cpu_ref'] = {'error': repr(e)}
    return out


def cpu_ref_lines(pkg, n, seconds=4.0, batch=4096):
    """The stronger CPU line: the build's own C++ backend behind the same C ABI
    (lib2048_cpu.so, 2048_amd/csrc/cpu_ref.cpp — the kernels' integer headers compiled for the host; explicit backend, not a
    fallback and not the oracle), the same synchronous TD(0) step on `batch` lanes, 1 thread and all the host threads this process may use (at most 16: the CPU share of a 1-GPU box)."""
    lines = {'what': f'lib2048_cpu.so (csrc/cpu_ref.cpp), full TD(0) step, n={n}, {batch} lanes, sum rule, ~{seconds:.0f} s each', 'unit': 'board-steps/s'}
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    allt = max(1, min(usable, 16))       # (a 1-GPU box gives this process a share of 16 cores whatever the host's core count says)
    for label, threads, lanes in (('one_thread', 1, batch), ('all_threads', allt, batch * 16)):
        os.environ['G2048_CPU_THREADS'] = str(threads)
        eng = pkg.Engine(lanes, n=n, seed=2048, backend='cpu')
        eng.init_weights(seed=7, scale=0.01)
        alpha = 0.25 * NUM_FEAT[n] / (8.0 * lanes)
        eng.td_steps(alpha, 8)
        steps, chunk = 0, 8
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            eng.td_steps(alpha, chunk)
            steps += chunk
        dt = time.perf_counter() - t0
        eng.close()
        lines[label] = {'value': steps * lanes / dt, 'threads': threads, 'lanes': lanes}
    os.environ.pop('G2048_CPU_THREADS', None)
    return lines


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
    elif args.workload == 'lookahead':
        # Game.look_forward (game_logic.py:214-243) for many positions at once: every tree expanded, evaluated and reduced on the
        # device (csrc/lookahead.hip).  since_empty = 17 makes every node an inner node (no early leaves), so a call is exactly
        # roots x (4 width)^depth leaf slots; the boards travel over PCIe once per call (16 B per root against ~100 B x leaves).
        B = 65536 if args.batch == 1 << 20 else args.batch
        n, depth, width = args.n_tuple, 2, 4
        eng = pkg.Engine(B, n=n, seed=2048)
        eng.step_random(120)
        eng.init_weights(seed=7, scale=0.01)
        boards = eng.get_boards()
        after, _, changed = eng.boards_move_all(boards)
        roots = np.ascontiguousarray(after[np.arange(B), np.argmax((changed[:, None] >> np.arange(4)[None, :]) & 1, axis=1)])   # one afterstate per lane
        for _ in range(max(1, args.warmup // 8)):
            eng.boards_look_forward(roots, depth, width, 17)
        t0 = time.perf_counter()
        reps = max(1, args.steps // 8)
        for _ in range(reps):
            v = eng.boards_look_forward(roots, depth, width, 17)
        dt = (time.perf_counter() - t0) / reps
        leaves = B * (4 * width) ** depth
        by = 16 + 4 * NUM_FEAT[n] + 4 + 2                   # a leaf slot: board + F gathers + value + valid / kind bytes
        out = dict(workload=f'look-ahead: V_{depth}(board) of {B} positions, width {width}, every node expanded, {n}-tuple table (g2048_boards_look_forward)',
                   batch=B, steps=reps, value=B / dt, unit='positions/s', leaf_slots_per_s=leaves / dt, ms_per_call=dt * 1e3,
                   algorithmic_GBps=by * leaves / dt / 1e9, finite=bool(np.isfinite(v).all()))
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


def lib_sha256():
    import hashlib
    with open(os.path.join(ROOT, '2048_amd', 'lib2048_hip.so'), 'rb') as f:
        return hashlib.sha256(f.read()).hexdigest()


def load_traffic(n, B):
    """PMC-measured per-launch counters of the kernels (tools/pmc_traffic.py -> profiles/traffic.json), valid only for the
    library they were measured on: the file carries the sha256 of lib2048_hip.so and a stale stamp yields nothing."""
    path = os.path.join(ROOT, 'profiles', 'traffic.json')
    if not os.path.exists(path):
        return None, 'profiles/traffic.json absent'
    with open(path) as f:
        t = json.load(f)
    if t.get('so_sha256') != lib_sha256():
        return None, 'profiles/traffic.json was measured on another build of lib2048_hip.so (stale stamp): rerun tools/r03_profiles.sh'
    entry = t.get('entries', {}).get(f'n{n}_b{B}')
    return entry, (None if entry else f'no PMC entry for n={n}, {B} lanes')


def build_roofline(eng, n, B, ms_step, ms_play, ms_owner, ms_tail, ms_apply, st, copy_gbps):
    """The measurement block of the JSON line.  Top level: the contract's fields for the dominant
    kernel — `achieved` = the reference algorithm's bytes for that kernel over its measured duration, against the
    8 TB/s HBM peak, `traffic` = its PMC-measured HBM bytes.  Below it, per kernel, the bound each one actually hits, every
    fraction <= 1 by construction (bytes or operations the kernel really moves / performs over its own time):
      k_td_play          L2 -> L1 request bytes (TCP_TCC_READ_REQ x 64 B, PMC) against the L2 peak; VALU-busy share (PMC)
      k_td_update_owner  record bytes its workgroups scan (from the plan in force) against the L2 peak; LDS atomic adds against
                         the measured ds_add_u64 rate
      whole step         PMC-measured HBM bytes against 8 TB/s, and the implementation's compulsory bytes (implementation_bytes)"""
    by_play, by_update = algorithmic_bytes(n)
    im_play, im_update = implementation_bytes(n)
    play_name = f'k_td_play<{n}>'
    kernels_ms = {play_name: ms_play, f'k_td_update_owner<{n}>': ms_owner, 'k_apply_orbits': ms_apply}
    if n == 6:
        kernels_ms['k_hex_* (f_6 orbits: count, plan, scatter, owner)'] = ms_tail
    dominant = max(kernels_ms, key=kernels_ms.get)
    pmc, pmc_note = load_traffic(n, B)
    pmc = pmc or {}

    def pm(kernel, counter):
        v = pmc.get(kernel, {}).get(counter)
        return float(v) if v is not None else None

    achieved = by_play * B / (ms_play * 1e-3) / 1e9
    # `bound`: what the dominant kernel is limited by, as the counters say: the request rate of the CU's L1
    # miss path (TCP -> L2, ~0.3 requests per cycle and CU), not HBM bandwidth.  `frac` stays what the contract defines —
    # reference-algorithm bytes over kernel time against the HBM peak — and is repeated under the name that says what it is.
    out = {'bound': 'l1-miss-path (L2 request rate of the table gathers); HBM is at `kernels.*.limits[bound=hbm].frac`', 'kernel': play_name,
           'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS, 'algorithmic_frac_of_hbm': achieved / HBM_PEAK_GBS,
           'traffic': pm('k_td_play', 'hbm_bytes'), 'traffic_source': pmc_note or 'profiles/traffic.json (tools/pmc_traffic.py), same lib2048_hip.so',
           'algorithmic_bytes_per_launch': by_play * B, 'ms_per_launch': ms_play, 'longest_kernel': dominant, 'ms_kernels': kernels_ms,
           'note': 'achieved = reference-algorithm bytes (SURVEY.md 8d: 72 + 4 F 4 + 20 per board-step) / kernel time: the contract\'s yardstick.  '
                   'The table gathers it counts are served by L2 / LDS, not HBM: the bounds the kernels really hit are under `kernels`.  '
                   'ms_per_launch is the HIP-event average of 20 steady-state steps of the TIMED path (sum rule, conditioned boards).  A rocprofv3 '
                   'per-kernel average over this whole command also contains the young boards of the conditioning phase (faster) and, unless '
                   '--no-mean-line / --trained-steps 0 are given, the mean-rule leg and the trained-agent leg (slower: `trained_agent.ms_kernels`); '
                   'profiles/r04_bench_driver_timed_path_kernel_stats.csv is the trace of the timed path alone.'}
    kern = {}
    # ---- k_td_play
    k = {'ms': ms_play, 'share_of_step': ms_play / ms_step, 'limits': []}
    req = pm('k_td_play', 'TCP_TCC_READ_REQ_sum')
    if req is not None:
        l2 = req * 64.0 / (ms_play * 1e-3) / 1e9
        k['limits'].append({'bound': 'l2', 'what': 'L2 -> L1 read requests x 64 B (PMC TCP_TCC_READ_REQ_sum)', 'bytes_per_launch': req * 64.0,
                            'achieved': l2, 'peak': L2_PEAK_GBS, 'unit': 'GB/s', 'frac': l2 / L2_PEAK_GBS})
    valu, cyc = pm('k_td_play', 'SQ_ACTIVE_INST_VALU'), pm('k_td_play', 'GRBM_GUI_ACTIVE')
    if valu is not None and cyc:
        # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the chip's 1 024 SIMDs (MI355X_MICROARCH.md, PMC units);
        # GRBM_GUI_ACTIVE is the kernel's busy cycles summed over the 8 XCDs
        busy = (valu * 4.0 / 1024.0) / (cyc / 8.0)
        k['limits'].append({'bound': 'valu', 'what': '(SQ_ACTIVE_INST_VALU x 4 / 1 024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs): share of the kernel\'s cycles in which a SIMD issues VALU work',
                            'achieved': busy, 'peak': 1.0, 'unit': 'busy fraction', 'frac': busy})
    hb = pm('k_td_play', 'hbm_bytes')
    if hb is not None:
        g = hb / (ms_play * 1e-3) / 1e9
        k['limits'].append({'bound': 'hbm', 'what': 'PMC FETCH_SIZE + WRITE_SIZE (corrected, tools/pmc_traffic.py)', 'bytes_per_launch': hb, 'achieved': g,
                            'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': g / HBM_PEAK_GBS})
    k['compulsory_hbm_bytes_per_launch'] = im_play * B
    kern[play_name] = k
    # ---- k_td_update_owner: what its plan makes it scan
    k = {'ms': ms_owner, 'share_of_step': ms_owner / ms_step, 'limits': []}
    try:
        plan = eng.debug_owner_plan()
        rec_bytes = {0: 12, 1: 12, 2: 12, 3: 12, 4: 6, 5: 20} if n >= 4 else {}
        scans = {}
        for row in plan:
            scans[(int(row[0]), int(row[1]))] = 1
        scanned = sum((rec_bytes.get(v, 20) if n >= 4 else 20) * B for (v, _) in scans)
        if scanned and ms_owner > 0:
            g = scanned / (ms_owner * 1e-3) / 1e9
            k['limits'].append({'bound': 'l2', 'what': f'{len(scans)} chunk scans of the step\'s records (8 + 4, 16 + 4 or 2 + 4 bytes per record and scan), L2 / Infinity-Cache resident',
                                'bytes_per_launch': scanned, 'achieved': g, 'peak': L2_PEAK_GBS, 'unit': 'GB/s', 'frac': g / L2_PEAK_GBS})
    except Exception as e:
        k['plan_error'] = repr(e)
    if ms_owner > 0:
        adds = ORBIT_ADDS[n] * B * 1.0
        rate = adds / (ms_owner * 1e-3)
        k['limits'].append({'bound': 'lds-atomic', 'what': f'{ORBIT_ADDS[n]} ds_add_u64 per record (orbit + coset reduced; every lane has a record in steady state)',
                            'adds_per_launch': adds, 'achieved': rate, 'peak': LDS_ATOMIC_PEAK, 'unit': 'adds/s', 'frac': rate / LDS_ATOMIC_PEAK})
    hb = pm('k_td_update_owner', 'hbm_bytes')
    if hb is not None:
        g = hb / (ms_owner * 1e-3) / 1e9
        k['limits'].append({'bound': 'hbm', 'what': 'PMC FETCH_SIZE + WRITE_SIZE', 'bytes_per_launch': hb, 'achieved': g, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                            'frac': g / HBM_PEAK_GBS})
    k['compulsory_hbm_bytes_per_launch'] = im_update * B
    kern[f'k_td_update_owner<{n}>'] = k
    # ---- k_apply_orbits: two streams over the orbit tables (read this step's, clear the next one's) + the touched table entries
    k = {'ms': ms_apply, 'share_of_step': ms_apply / ms_step, 'limits': []}
    hb = pm('k_apply_orbits', 'hbm_bytes')
    if hb is not None and ms_apply > 0:
        g = hb / (ms_apply * 1e-3) / 1e9
        k['limits'].append({'bound': 'hbm', 'what': 'PMC FETCH_SIZE + WRITE_SIZE', 'bytes_per_launch': hb, 'achieved': g, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                            'frac': g / HBM_PEAK_GBS})
    kern['k_apply_orbits'] = k
    out['kernels'] = kern
    # ---- whole step
    step_hbm = None
    if pmc:
        vals = [v.get('hbm_bytes') for v in pmc.values() if isinstance(v, dict) and v.get('hbm_bytes') is not None]
        step_hbm = float(sum(vals)) if vals else None
    comp = (im_play + im_update) * B
    ws = {'ms': ms_step, 'compulsory_hbm_bytes': comp, 'compulsory_GBps': comp / (ms_step * 1e-3) / 1e9,
          'compulsory_frac_of_hbm_peak': comp / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
          'reference_algorithm_bytes': (by_play + by_update) * B,
          'note': 'compulsory = what this implementation must stream per step when only the table is cache-resident (implementation_bytes: '
                  f'{im_play} + {im_update} B per lane); reference_algorithm_bytes (SURVEY.md 8d, {by_play + by_update} B per board-step) is the reference\'s '
                  'gather + 8-image read-modify-write count, most of which this implementation turns into LDS adds: quoted for the record, not as a rate'}
    if copy_gbps:
        # the headline distance to the roofline: the step's time over the time its compulsory bytes would take at the box's own
        # measured copy rate (and at the 8 TB/s datasheet peak)
        ws['hbm_floor_ms_at_measured_copy_rate'] = comp / (copy_gbps * 1e9) * 1e3
        ws['over_hbm_floor'] = ms_step / ws['hbm_floor_ms_at_measured_copy_rate']
    ws['over_hbm_floor_at_peak'] = ms_step / (comp / (HBM_PEAK_GBS * 1e9) * 1e3)
    if step_hbm is not None:
        ws.update({'hbm_bytes_measured': step_hbm, 'hbm_GBps': step_hbm / (ms_step * 1e-3) / 1e9,
                   'frac_of_hbm_peak': step_hbm / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS})
    out['whole_step'] = ws
    out['measured_copy_GBps'] = copy_gbps
    fr = [out['frac'], ws['compulsory_frac_of_hbm_peak']] + [lim['frac'] for kk in kern.values() for lim in kk['limits'] if 'frac' in lim]
    if any(f > 1.0 for f in fr):
        out['invalid'] = 'a fraction above 1: bookkeeping error'
    return out


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
    ap.add_argument('--trained-steps', type=int, default=3000,
                    help='after the mean-rule leg: train this many more steps under the mean rule, then time K steps on the trained agent\'s boards (0: skip)')
    ap.add_argument('--no-mean-line', action='store_true', help='skip the secondary measurement of the per-slot mean rule')
    ap.add_argument('--workload', default='td', choices=['td', 'env', 'eval', 'lookahead'],
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
    ap.add_argument('--comm-algo', default='', choices=['', 'allreduce', 'rsag'],
                    help='native exchange: one ncclAllReduce (default) or ncclReduceScatter + ncclAllGather (G2048_COMM_ALGO=rsag: all xGMI links busy by construction)')
    ap.add_argument('--deadline', type=float, default=900.0,
                    help='self-launched N > 1 runs: seconds after which the parent terminates every rank and fails')
    ap.add_argument('--fault-inject', default='', help=argparse.SUPPRESS)
    ap.add_argument('--backend', default='nccl', help='torch.distributed backend for control traffic (nccl = RCCL; gloo to rehearse on one GPU)')
    args = ap.parse_args()

    if args.comm_algo:
        os.environ['G2048_COMM_ALGO'] = args.comm_algo          # (read by g2048_create; inherited by self-launched ranks)
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
            else:
                local_rank = 0            # no GPU: only an explicit G2048_BACKEND=cpu gets further (the host is its device 0)
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
            if torch.cuda.is_available():
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
    ms_step = dt / K * 1e3
    roofline = build_roofline(eng, n, B, ms_step, ms_play, ms_owner, ms_tail, ms_apply, st, copy_gbps)

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
        comm = {'kind': comm_kind + (' as ncclReduceScatter + ncclAllGather' if os.environ.get('G2048_COMM_ALGO') == 'rsag' and 'native' in comm_kind else ''), 'rank_seen': seen[0], 'nranks_seen': seen[1], 'nranks_expected': world,
                'payload_bytes': payload, 'epoch_steps': args.epoch, 'exchanges_per_timed_region': -(-K // args.epoch),
                'allreduce_plus_apply_ms': statistics.median(xs), 'allreduce_plus_apply_ms_repeats': xs,
                'predicted_allreduce_ms': {'ring_one_link': 2 * (world - 1) / world * payload / link * 1e3,
                                           'all_links_direct': (2 * payload / world / link * 1e3) if world > 1 else 0.0}}
        if seen[1] != world:
            raise SystemExit(f'[bench] the communicator reports {seen[1]} ranks, expected {world}')
        # what every rank ran on, gathered: its lane shard (lane0 = rank x B: episodes are sharded, nothing else is) and two
        # checksums of its table after the last exchange — the replicas must be identical
        import torch
        wflat = eng.get_weights().astype(np.float64)
        mine = [float(rank * B), float(wflat.sum()), float(np.dot(wflat, wflat))]
        del wflat
        box = torch.zeros((world, 3), dtype=torch.float64, device='cuda' if args.backend == 'nccl' else 'cpu')
        box[rank] = torch.tensor(mine, dtype=torch.float64)
        dist.all_reduce(box)
        box = box.cpu().numpy()
        comm['lane0_per_rank'] = [int(x) for x in box[:, 0]]
        comm['table_checksums_per_rank'] = [[float(a), float(b)] for a, b in box[:, 1:]]
        comm['replicas_identical'] = bool((box[:, 1:] == box[0, 1:]).all())
        if not comm['replicas_identical'] or len(set(comm['lane0_per_rank'])) != world:
            raise SystemExit(f'[bench] replicas differ or lane shards overlap: {comm["lane0_per_rank"]} {comm["table_checksums_per_rank"]}')

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

    trained = None
    if mean_line is not None and args.trained_steps > 0:
        # the number a user of QAgent.train_run sees: the per-slot mean rule on the boards of an agent that has learned for
        # `--trained-steps` steps in this very run (bigger tiles: colder gathers, more busy chunks); same lanes, same kernels
        eng.stats_reset()
        t0 = time.perf_counter()
        eng.td_steps(args.alpha, args.trained_steps)
        eng.sync()
        train_s = time.perf_counter() - t0
        tt = []
        for _ in range(max(1, args.repeats)):
            eng.timer_start()
            eng.td_steps(args.alpha, K)
            tt.append(eng.timer_stop())
        tm = statistics.median(tt)
        ts = eng.stats()
        kp, ko, kt, ka = eng.td_steps_kernel_ms(args.alpha, 20)
        trained = {'update_rule': 'mean', 'trained_steps': args.trained_steps, 'trained_seconds': train_s, 'episodes_while_training': ts['episodes'],
                   'mean_score_while_training': ts['score_sum'] / max(1, ts['episodes']), 'value': B * K / (tm * 1e-3), 'unit': 'board-steps/s',
                   'ms_per_step': tm / K, 'ms_kernels': {'k_td_play': kp, 'k_td_update_owner': ko, 'k_apply_orbits_mean': ka}}

    if rank == 0:
        out = {
            'metric': 'board-steps/sec at batch 2^20 (full TD(0) step: move x4, n-tuple gather, greedy select, 8-symmetry scatter-add, spawn)',
            'value': world * B * K / dt,
            'unit': 'board-steps/s',
            'n_gpus': world, 'steps': K, 'warmup': W,
            'ms_per_step': ms_step,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'u8 boards / int32 scores / f32 weights', 'data': 'synthetic',
            'config': {'workload': (f'BASELINE config 5: {world} x MI355X data-parallel TD(0) training, 6-tuple table ({eng.slots * 4} B), ' if n == 6 and world > 1 else
                                    f'BASELINE config 4: full TD(0) loop, {n}-tuple table ({eng.slots * 4} B), ')
                                   + f'{B} concurrent episodes per GPU, auto-reset',
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
        if trained:
            out['trained_agent'] = trained
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
