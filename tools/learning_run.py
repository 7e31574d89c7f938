#!/usr/bin/env python3
"""Learning evidence: batched TD(0) on the device through the reference-shaped API; prints the training log and a
greedy trial (cf. the quality tables of the reference, README.md:79-126)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from game2048.r_learning import *   # noqa
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 16
episodes = int(sys.argv[3]) if len(sys.argv) > 3 else 5_000_000
alpha = float(sys.argv[4]) if len(sys.argv) > 4 else 0.25
rule = os.environ.get('G2048_RULE', 'mean')
agent = QAgent(name=f'a{n}', storage='local', console='local', n=n, alpha=alpha, batch=batch, seed=1, decay_step=episodes // 4, rule=rule)
lines = []
def log(x):
    x = str(x)
    if 'average over last' in x or 'reached in' in x or 'learning rate' in x or 'Total time' in x:
        print(x.strip(), flush=True)
agent.print = log
t0 = time.time()
agent.train_run(num_eps=episodes, saving=False)
st = agent.engine.stats()
print(f'trained {agent.step} episodes ({st["moves"]} board-steps) in {time.time() - t0:.1f} s; top score {agent.top_score}', flush=True)
res = QAgent.trial(estimator=agent.evaluate, num=1000, storage='local', console='local')
