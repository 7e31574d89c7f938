#!/usr/bin/env python3
"""Diagnostic: update-kernel time under a fixed plan policy, with and without adds (alpha = 0 -> scan only)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module('2048_amd')
n, B = 5, 1 << 20
eng = pkg.Engine(B, n=n, seed=2048)
eng.init_weights(seed=7, scale=0.01)
alpha = 0.25 * eng.num_feat / (8.0 * B)
eng.td_steps(alpha, 300)          # desynchronise the games a little, let the planner settle
for rep in range(3):
    a, b = eng.td_steps_profiled(alpha, 10)
    a0, b0 = eng.td_steps_profiled(0.0, 10)
    print(f'steady: play {a:.3f} update {b:.3f} ms | alpha=0 (scan only): play {a0:.3f} update {b0:.3f} ms', flush=True)
