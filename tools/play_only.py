#!/usr/bin/env python3
"""A short run of the bench workload for profilers: N steps at 2^20 lanes, n = 5 (env EVERY = lane sort period)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module('2048_amd')
B = 1 << 20
eng = pkg.Engine(B, n=5, seed=2048)
eng.init_weights(seed=7, scale=0.01)
eng.set_lane_sort(int(os.environ.get('EVERY', 16)))
eng.td_steps(0.25 * 21 / (8.0 * B), int(os.environ.get('STEPS', 80)))
eng.sync()
