"""Mean time per wave and phase of k_td_play, and how many of its workgroups run at once.  Needs the instrumented build
(-DG2048_EXP_PHASES, optionally with -DG2048_EXP_NOGATHER; pass it with G2048_LIB): wall-clock stamps at the phase boundaries,
summed per wave in a __device__ array, and start / end stamps of every workgroup (entry point g2048_debug_phases)."""
import ctypes, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module('2048_amd')
lib = importlib.import_module('2048_amd._lib').load()
n, B = 5, 1 << 20
eng = pkg.Engine(B, n=n, seed=2048)
eng.init_weights(seed=7, scale=0.01)
eng.set_lane_sort(int(os.environ.get('EVERY', 0)))
alpha = 0.25 * eng.num_feat / (8.0 * B)
eng.td_steps(alpha, 320)
eng.sync()
out = (ctypes.c_ulonglong * 8)()
spans = np.zeros(2 * 8192, np.uint64)
lib.g2048_debug_phases(out, None, 1)
steps = 20
k = eng.td_steps_kernel_ms(alpha, steps)
eng.sync()
lib.g2048_debug_phases(out, spans.ctypes.data_as(ctypes.c_void_p), 0)
waves = B / 64 * steps
names = ['block hand-out (barrier, counter)', 'state loads + moves', 'features, gathers, sums, select', 'pick, spawn, terminal, orbit idx', 'dw max + move counts (wave reductions)',
         'leaving the loop + flush', 'finished games: statistics, new game', 'stores (issue)']
print(f'k_td_play {k[0] * 1e3:.1f} us; mean microseconds per wave and 64-lane block, by phase:')
tot = 0
for j in range(8):
    us = out[j] / waves / 100.0
    tot += us
    print(f'  {names[j]:36s} {us:8.2f} us')
print(f'  total per wave and block {tot:.2f} us  ->  x {B // 64} wave-blocks / kernel time = {tot * (B // 64) / (k[0] * 1e3):.0f} waves busy on average')
t = spans.reshape(8192, 2).astype(np.int64)
t = t[(t[:, 0] > 0) & (t[:, 1] > t[:, 0])]
t = t[t[:, 0] > t[:, 0].max() - 200000]            # the last launch only (stamps of earlier launches linger for unused slots)
t0 = t[:, 0].min()
st, en = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0
ev = sorted([(x, 1) for x in st] + [(x, -1) for x in en])
cur = mx = 0
for _, d in ev:
    cur += d
    mx = max(mx, cur)
print(f'last launch: {len(t)} workgroups; starts: {np.sum(st < 2)} within 2 us, {np.sum(st < 10)} within 10 us, {np.sum(st < 50)} within 50 us, last at {st.max():.1f} us; '
      f'ends between {en.min():.1f} and {en.max():.1f} us; mean duration {(en - st).mean():.1f} us; most at once {mx}')
