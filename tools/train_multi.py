#!/usr/bin/env python3
"""Multi-GPU training through the reference's own surface: QAgent.train_run on every rank of a torch.distributed job.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 \\
        tools/train_multi.py --n 5 --batch 262144 --episodes 4000000 --name A5

One process per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the launcher).  Every rank builds the same agent;
rank r plays lanes [r * batch, (r + 1) * batch) of the job's world * batch concurrent episodes; the accumulated weight
deltas are all-reduced over RCCL every --epoch board-steps (2048_amd/parallel.py); rank 0 prints the reference's
training log and saves the agent (r_learning.py:269-346).  --backend gloo rehearses the same job on one GPU."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=5)
    ap.add_argument('--batch', type=int, default=1 << 18, help='concurrent episodes per GPU')
    ap.add_argument('--episodes', type=int, default=1000000)
    ap.add_argument('--alpha', type=float, default=0.25)
    ap.add_argument('--epoch', type=int, default=64)
    ap.add_argument('--rule', default=None, choices=[None, 'sum', 'mean'])
    ap.add_argument('--name', default='agent_multi')
    ap.add_argument('--backend', default='nccl')
    ap.add_argument('--comm', default='native', choices=['native', 'torch'])
    ap.add_argument('--save', action='store_true')
    ap.add_argument('--dump', default=None, help='write a summary of this rank (npz) to DUMP.<rank>.npz')
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    rank, local = int(os.environ.get('RANK', 0)), int(os.environ.get('LOCAL_RANK', 0))
    if args.backend == 'nccl':
        torch.cuda.set_device(local)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    else:
        if torch.cuda.device_count():
            local = local % torch.cuda.device_count()
            torch.cuda.set_device(local)
        else:
            local = 0                     # no GPU: the job runs on the CPU backend if G2048_BACKEND=cpu says so (explicitly)
        dist.init_process_group(args.backend)
    import game2048.r_learning as rl
    np.random.seed(2048)
    agent = rl.QAgent(name=args.name, storage='local', console='local', n=args.n, alpha=args.alpha, batch=args.batch, device=local,
                      rule=args.rule, dist=dist, epoch=args.epoch, comm=args.comm)
    agent.train_run(num_eps=args.episodes, saving=args.save)
    if args.dump:
        w = agent.engine.get_weights()
        np.savez(f'{args.dump}.{rank}.npz', step=agent.step, alpha=agent.alpha, top_tile=agent.top_tile, top_score=agent.top_score,
                 history=np.array(agent.train_history), wsum=float(w.astype(np.float64).sum()), w_head=w[:4096], w_tail=w[-4096:],
                 sync=type(agent._sync).__name__, reduces=agent._sync.reduces if agent._sync else 0,
                 top_game_score=agent.top_game.score if agent.top_game is not None else -1)
    dist.barrier()
    if hasattr(agent._sync, 'close'):
        agent._sync.close()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
