#!/usr/bin/env python3
"""k_td_play / update leg times as the agent ages (bench workload, sum rule): the boards' tile distribution drifts."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module('2048_amd')
n, B = int(os.environ.get('N', 5)), 1 << 20
eng = pkg.Engine(B, n=n, seed=2048)
eng.init_weights(seed=7, scale=0.01)
alpha = 0.25 * eng.num_feat / (8.0 * B)
done = 0
for upto in [int(x) for x in os.environ.get("UPTO", "64,264,1000,2500,5000,10000").split(",")]:
    eng.td_steps(alpha, upto - done)
    done = upto
    a, b = eng.td_steps_profiled(alpha, 16)
    done += 16
    st = eng.stats()
    plan = eng.debug_owner_plan()
    print(f'after {done:6d} steps: play {a*1e3:6.1f} us  update {b*1e3:6.1f} us  workgroups {len(plan)} over {len(set(plan[:,1].tolist()))} chunks  mean score {st["score_sum"]/max(1,st["episodes"]):.0f}', flush=True)
    eng.stats_reset()
