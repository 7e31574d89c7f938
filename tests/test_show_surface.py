"""BASELINE config 1 / north_star: "keeps the reference's Game and Q_agent Python surface so show.py still drives it".

tests/golden/show_transcript.npz is what the REFERENCE's own show.py painted when tests/golden/make_show_transcript.py ran it
headless (options 1, 2, 3; scripted input; tests/pygame_stub.py as pygame) on top of this repository's game2048 package on the
CPU backend.  Here the same sequence of surface calls (tests/show_driver.drive: input_name -> load_s3 / QAgent.load_agent ->
QAgent.trial(num=100) -> the replay / watch loops) runs again and must paint the same frames:
  * on lib2048_cpu.so (-m "not gpu"), which also proves that show_driver restates show.py's call sequence faithfully;
  * on lib2048_hip.so (-m gpu): the HIP kernels behind the surface give show.py the same boards, moves, scores and games;
  * and, where the reference is present (the build container), show.py itself is run again and compared (--check).
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import show_driver

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _compare(rl, golden, tmp_path):
    g = golden('show_transcript.npz')
    names = show_driver.prepare_storage(rl, tmp_path / 'store')
    assert names == str(g['names']).split('\n')
    for option in (1, 2, 3):
        frames, scores = show_driver.drive(rl, option)
        want = g[f'opt{option}_frames']
        got = np.array(frames, np.int64)
        assert got.shape == want.shape, (option, got.shape, want.shape)
        bad = np.nonzero((got != want).any(axis=1))[0]
        assert len(bad) == 0, (option, int(bad[0]), got[bad[0]].tolist(), want[bad[0]].tolist())
        if option == 2:
            assert sorted(scores, reverse=True) == g['opt2_scores'].tolist()
        assert want[-1, 3] == 1 and (want[:-1, 3] == 0).all()          # one "Over!" frame, at the end


def test_show_call_sequence_on_the_cpu_backend(golden, tmp_path, monkeypatch):
    monkeypatch.setenv('G2048_BACKEND', 'cpu')
    monkeypatch.setenv('G2048_CPU_THREADS', '4')
    import game2048.r_learning as rl
    _compare(rl, golden, tmp_path)


@pytest.mark.skipif(not os.path.exists('/root/reference/show.py'), reason='the reference is only present in the build container')
def test_reference_show_py_itself_drives_the_surface():
    """The reference's show.py, executed as __main__ with scripted input, against game2048/ -> 2048_amd/ on the CPU backend."""
    env = dict(os.environ, G2048_BACKEND='cpu', G2048_CPU_THREADS='4')
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'golden', 'make_show_transcript.py'), '--check'], capture_output=True, text=True,
                         timeout=600, env=env)
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-1500:]
    assert 'identical to show_transcript.npz' in res.stdout


@pytest.mark.gpu
def test_show_call_sequence_on_the_hip_backend(golden, tmp_path, monkeypatch):
    monkeypatch.setenv('G2048_BACKEND', 'hip')
    import game2048.r_learning as rl
    _compare(rl, golden, tmp_path)
