#!/usr/bin/env python3
"""Interleaved A/B timing of environment knobs (or library builds) on the bench workload.

    python3 tools/ab.py "base" "G2048_SORT_EVERY=32" "G2048_SORT_EVERY=64" "G2048_LIB=/path/lib.so G2048_PLAY_DYNAMIC=3"

Every configuration runs REPS times (default 4) in a fresh process each (the knobs are read at g2048_create, the library at
import), the configurations taking turns so that drift of the box hits all of them alike.  Prints mean and spread of the
step time and of the per-kernel times.  Identical builds differ by up to 2 % from run to run on one box: that is the noise
floor of anything this prints."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import importlib, json, os, sys
sys.path.insert(0, %r)
pkg = importlib.import_module('2048_amd')
n, B = int(os.environ.get('N', 5)), int(os.environ.get('B', 1 << 20))
eng = pkg.Engine(B, n=n, seed=2048)
eng.init_weights(seed=7, scale=0.01)
alpha = 0.25 * eng.num_feat / (8.0 * B)
if os.environ.get('RULE', 'sum') == 'mean':
    eng.set_update_rule(1)
    alpha = 0.25
eng.td_steps(alpha, int(os.environ.get('AGE', 320)))
eng.sync()
ts = []
for rep in range(3):
    eng.timer_start()
    eng.td_steps(alpha, 192)
    ts.append(eng.timer_stop() / 192)
k = eng.td_steps_kernel_ms(alpha, 32)
print(json.dumps({'ms': sorted(ts)[1], 'k': [float(x) for x in k]}))
''' % ROOT


def main():
    configs = sys.argv[1:] or ['base']
    reps = int(os.environ.get('REPS', 4))
    res = {c: [] for c in configs}
    for r in range(reps):
        for c in configs:
            env = dict(os.environ)
            if c != 'base':
                for kv in c.split():
                    k, v = kv.split('=', 1)
                    env[k] = v
            out = subprocess.run([sys.executable, '-c', CHILD], env=env, capture_output=True, text=True, timeout=300)
            line = [l for l in out.stdout.splitlines() if l.startswith('{')]
            if not line:
                print(f'{c}: run failed: {out.stderr[-300:]}', flush=True)
                continue
            res[c].append(json.loads(line[-1]))
    base = None
    for c in configs:
        if not res[c]:
            continue
        ms = np.array([x['ms'] for x in res[c]])
        k = np.array([x['k'] for x in res[c]])
        base = base if base is not None else ms.mean()
        print(f'{c:60s} ms/step {ms.mean():.4f} +- {ms.std():.4f} ({100 * (ms.mean() / base - 1):+5.1f} %)   play {k[:, 0].mean():.4f} owner {k[:, 1].mean():.4f} '
              f'tail {k[:, 2].mean():.4f} apply {k[:, 3].mean():.4f}', flush=True)


if __name__ == '__main__':
    main()
