"""Start / end clock of every k_td_play workgroup (how many run at once).  Needs an instrumented build (not kept in the
tree) with a __device__ span array and g2048_debug_spans(out); results are quoted in DESIGN.md section 4."""
import importlib, os, sys, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module('2048_amd')
lib = importlib.import_module('2048_amd._lib').load()
n, B = int(os.environ.get('N', 5)), 1 << 20
eng = pkg.Engine(B, n=n, seed=2048)
eng.init_weights(seed=7, scale=0.01)
alpha = 0.25 * eng.num_feat / (8.0 * B)
eng.td_steps(alpha, 300); eng.sync()
a, b = eng.td_steps_profiled(alpha, 5); eng.sync()
out = np.zeros(8192 * 2, np.uint64)
lib.g2048_debug_spans(out.ctypes.data_as(ctypes.c_void_p))
g = int(os.environ.get('G2048_PLAY_WGS', 768))
g = min(g, 4096)
t = out[:2 * g].reshape(g, 2).astype(np.int64)
t0 = t[:, 0].min()
st, en = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0
ev = sorted([(x, 1) for x in st] + [(x, -1) for x in en])
cur = mx = 0
for _, d in ev:
    cur += d; mx = max(mx, cur)
print(f'n={n} grid {g}: play {a*1e3:.1f} us; kernel span {en.max():.1f} us; max concurrent workgroups {mx}; '
      f'starts: {np.sum(st < 5)} within 5 us, {np.sum(st < 50)} within 50 us; mean duration {(en - st).mean():.1f} us')
