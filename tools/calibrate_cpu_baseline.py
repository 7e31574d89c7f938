#!/usr/bin/env python3
"""Calibrate the CPU baselines on ONE machine ("cpu_baseline is uncalibrated").

The reference cannot travel to the GPU box, so bench.py times the oracle's reference-structured port there
(oracle/ref_scalar.py).  This script — run in the build container, where /root/reference can be imported — times, on the
SAME core and the same workload (whole TD(0) episodes from fresh random weights, spawn draws from the device RNG spec):

  1. the imported reference itself      QAgent.episode()                 (r_learning.py:224-252)
  2. the oracle's port                  oracle.ref_scalar.Agent.episode  (what bench.py's cpu_baseline times)
  3. the build's C++ backend            lib2048_cpu.so, batch 1 and batch 4096, 1 thread

and rewrites the section "## 4. CPU baselines calibrated on one machine" of BASELINE.md with the numbers and ratios.

    python tools/calibrate_cpu_baseline.py [--seconds 10]
"""
import argparse
import importlib
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.golden.make_golden import DrawShim, import_reference  # noqa: E402

MARK = '## 4. CPU baselines calibrated on one machine'


def time_reference(gl, rl, n, seconds):
    np.random.seed(0)
    agent = rl.QAgent(name='cal', storage='local', console='local', n=n, alpha=0.25)
    gl.random = DrawShim(2048, 0)
    moves = games = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        g = agent.episode()
        moves += g.odometer
        games += 1
    return moves / (time.perf_counter() - t0), games


def time_port(n, seconds):
    from oracle import ref_scalar as rs
    pkg = importlib.import_module('2048_amd')
    rs.row_table()
    np.random.seed(0)
    agent = rs.Agent(n=n, alpha=0.25)
    lane = pkg.rng.LaneRng(2048, 0)
    moves = games = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        _, _, m = agent.episode(lambda n_empty: pkg.rng.spawn_draw(lane.next(), n_empty))
        moves += m
        games += 1
    return moves / (time.perf_counter() - t0), games


def time_cpu_ref(n, batch, seconds, threads=1):
    os.environ['G2048_CPU_THREADS'] = str(threads)
    pkg = importlib.import_module('2048_amd')
    eng = pkg.Engine(batch, n=n, seed=2048, backend='cpu')
    eng.init_weights(seed=7, scale=0.01)
    alpha = 0.25 if batch == 1 else 0.25 * eng.num_feat / (8.0 * batch)
    chunk = 2000 if batch == 1 else 20
    eng.td_steps(alpha, chunk)
    steps = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        eng.td_steps(alpha, chunk)
        steps += chunk
    dt = time.perf_counter() - t0
    eng.close()
    return steps * batch / dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seconds', type=float, default=10.0)
    args = ap.parse_args()
    gl, rl = import_reference()
    import builtins
    real_print = builtins.print
    rows = []
    for n in (2, 4, 5):
        rl.print = lambda *a, **k: None
        ref, g1 = time_reference(gl, rl, n, args.seconds)
        port, g2 = time_port(n, args.seconds)
        c1 = time_cpu_ref(n, 1, min(args.seconds, 5.0))
        c4096 = time_cpu_ref(n, 4096, min(args.seconds, 5.0))
        rows.append((n, ref, port, port / ref, c1, c4096))
        real_print(f'n={n}: reference {ref:.0f}  port {port:.0f}  (ratio {port / ref:.2f})  cpu_ref batch 1 {c1:.0f}  batch 4096 {c4096:.0f} board-steps/s')
    cpu = open('/proc/cpuinfo').read()
    model = re.search(r'model name\s*:\s*(.*)', cpu)
    text = [MARK, '',
            f'`python tools/calibrate_cpu_baseline.py --seconds {args.seconds:g}` in the build container ({model.group(1).strip() if model else "unknown CPU"}, '
            f'{os.cpu_count()} logical cores visible, **1 used**; Python {sys.version.split()[0]}, NumPy {np.__version__}).  Same core, same workload for every column: '
            'whole TD(0) self-play episodes from fresh U[0, 0.01) weights, alpha 0.25, spawn draws from the device RNG spec.',
            '',
            '| n | imported reference `QAgent.episode` | oracle port (`oracle/ref_scalar.py`, what `bench.py` times as `cpu_baseline`) | port / reference | `lib2048_cpu.so` batch 1, 1 thread | `lib2048_cpu.so` batch 4096, 1 thread |',
            '|---|---:|---:|---:|---:|---:|']
    for n, ref, port, ratio, c1, c4096 in rows:
        text.append(f'| {n} | {ref:,.0f} | {port:,.0f} | {ratio:.2f} | {c1:,.0f} | {c4096:,.0f} |')
    text += ['',
             'board-steps/s.  The port is a restatement with the reference\'s own structures (dict row table, `np.rot90`, NumPy slicing encoders, '
             'list-of-lists float64 weights), so the ratio column is the calibration of the `cpu_baseline` figure in `BENCH_rNN.json`: '
             'reference-on-that-box ≈ port-on-that-box ÷ ratio.  Section 3\'s ±30 % sanity band is met when the ratio is within 0.7–1.3.', '']
    path = os.path.join(ROOT, 'BASELINE.md')
    body = open(path).read()
    if MARK in body:
        body = body[:body.index(MARK)].rstrip() + '\n\n'
    else:
        body = body.rstrip() + '\n\n'
    open(path, 'w').write(body + '\n'.join(text))
    real_print('BASELINE.md section 4 rewritten')


if __name__ == '__main__':
    main()
