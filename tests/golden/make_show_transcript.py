#!/usr/bin/env python3
"""BASELINE config 1: the REFERENCE's own show.py (abachurin/2048 @ /root/reference/show.py) driven headless against THIS
repository's `game2048` surface, in the build container (the reference's file stays where it is; nothing of it is copied).

    python tests/golden/make_show_transcript.py [--backend cpu|hip] [--check]

What happens: the repository root goes first on sys.path (so `from game2048.r_learning import *`, show.py:4, resolves to
game2048/ here -> 2048_amd/), tests/pygame_stub.py is registered as `pygame` / `pygame.locals`, `input` is scripted, and
show.py is executed as `__main__` (runpy) once per menu option:

    option 1   input_name('game') -> load_s3 -> input_speed -> Show().replay(game)                       (show.py:199-202)
    option 2   input_name('agent') -> QAgent.load_agent -> QAgent.trial(estimator=est, num=100, console='local')
               -> Show().replay(results[0])                                                              (show.py:204-212)
    option 3   the same agent -> Show().watch(estimator=est) to game over                                (show.py:213-216)

on the backend G2048_BACKEND names (default here: cpu — lib2048_cpu.so, the same C ABI without a GPU).  The store show.py's
menu lists is prepared by tests/show_driver.prepare_storage.  Every frame show.py paints is recorded by the stub and written
to show_transcript.npz: per option the frames [score, moves, move, over, 16 faces], the scores of option 2's hundred games,
and the lines show.py printed (timing lines dropped).  tests/test_show_surface.py replays the same call sequence
(tests/show_driver.drive) on the CPU backend and, under -m gpu, on the HIP backend, and compares with this file.
--check: run and compare with the stored file instead of writing it.
"""
import argparse
import builtins
import io
import os
import random
import runpy
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SHOW = '/root/reference/show.py'


def run_option(option, answers):
    """Execute show.py as __main__ with scripted input; returns (frames, printed lines, waits)."""
    from tests import pygame_stub, show_driver
    rec = pygame_stub.REC
    rec.reset()
    random.seed(show_driver.SEEDS[option])
    feed = iter(answers)
    asked = []

    def scripted_input(prompt=''):
        asked.append(prompt)
        return next(feed)
    out = io.StringIO()
    real_input, real_stdout = builtins.input, sys.stdout
    builtins.input, sys.stdout = scripted_input, out
    try:
        runpy.run_path(SHOW, run_name='__main__')
    except SystemExit:
        pass                                        # Show.replay leaves its closing loop through sys.exit() (show.py:123-124)
    finally:
        builtins.input, sys.stdout = real_input, real_stdout
    assert next(feed, None) is None, 'show.py did not ask for every scripted answer'
    lines = [ln for ln in out.getvalue().split('\n') if 'time' not in ln and 'shuffle' not in ln]
    return rec.frames(), lines, list(rec.waits), rec.quits_sent


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--backend', default='cpu')
    ap.add_argument('--check', action='store_true')
    args = ap.parse_args()
    os.environ['G2048_BACKEND'] = args.backend
    os.environ.setdefault('G2048_CPU_THREADS', '4')
    os.environ['S3_URL'] = 'none'
    sys.path[:] = [p for p in sys.path if not p.startswith('/root/reference')]
    sys.path.insert(0, ROOT)
    from tests import pygame_stub, show_driver
    pygame_stub.install()
    import game2048.r_learning as rl
    assert rl.__file__.startswith(ROOT), rl.__file__
    store = tempfile.mkdtemp(prefix='g2048_show_')
    names = show_driver.prepare_storage(rl, store)
    print('store:', names)
    agent_idx = next(i for i, v in enumerate(names) if v.startswith('a/'))
    game_idx = next(i for i, v in enumerate(names) if v.startswith('g/'))
    out = {}
    # option 1: a junk answer and a wrong index first (input_name / input_speed loop until they get a usable one), speed "0" = 50 ms
    frames, lines, waits, quits = run_option(1, ['1', 'zzz', '99', str(game_idx), 'fast', '0'])
    assert waits and set(waits) == {50} and quits >= 1
    out['opt1_frames'], out['opt1_lines'] = np.array(frames, np.int64), np.array('\n'.join(lines))
    frames, lines, waits, quits = run_option(2, ['2', str(agent_idx), '10'])
    assert set(waits) == {10}
    scores = [int(ln.split('result ')[1].split(',')[0]) for ln in lines if ln.startswith('game ') and 'result' in ln]
    assert len(scores) == 100
    out['opt2_frames'], out['opt2_scores'], out['opt2_lines'] = np.array(frames, np.int64), np.array(sorted(scores, reverse=True), np.int64), np.array('\n'.join(lines))
    frames, lines, waits, quits = run_option(3, ['3', str(agent_idx), '2000'])
    assert set(waits) == {2000}
    out['opt3_frames'], out['opt3_lines'] = np.array(frames, np.int64), np.array('\n'.join(lines))
    for k in (1, 2, 3):
        f = out[f'opt{k}_frames']
        print(f'option {k}: {len(f)} frames, final score {f[-1, 0]} after {f[-1, 1]} moves, over = {f[-1, 3]}')
    path = os.path.join(HERE, 'show_transcript.npz')
    if args.check:
        want = np.load(path)
        for k in out:
            assert np.array_equal(out[k], want[k]), k
        print(f'{args.backend}: identical to show_transcript.npz')
    else:
        np.savez_compressed(path, names=np.array('\n'.join(names)), **out)
        print(f'show_transcript.npz: {os.path.getsize(path) / 1024:.1f} KiB')


if __name__ == '__main__':
    main()
