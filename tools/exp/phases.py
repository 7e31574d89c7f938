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

# where do the slow workgroups sit?  (HW_ID of each workgroup's first wave: cu_id [11:8], sh_id [12], se_id [15:13]; XCC_ID [3:0])
if hasattr(lib, 'g2048_debug_phases_hw'):
    hw = np.zeros(8192, np.uint64)
    wave = np.zeros(8192 * 4 * 2, np.uint64)
    lib.g2048_debug_phases_hw(hw.ctypes.data_as(ctypes.c_void_p), wave.ctypes.data_as(ctypes.c_void_p))
    sp = spans.reshape(8192, 2).astype(np.int64)
    nwg = int(np.sum(sp[:, 1] > sp[:, 0]))
    nwg = min(nwg, 2048)
    sp, hw = sp[:nwg], hw[:nwg].astype(np.int64)
    W = int(os.environ.get('PLAY_TPB', 256)) // 64
    wv = wave.reshape(-1, 2).astype(np.int64)[:nwg * W].reshape(nwg, W, 2)
    t0 = sp[:, 0].min()
    end = (sp[:, 1] - t0) / 100.0
    xcc = (hw >> 32) & 0xF
    cu, sh, se = (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7
    blocks = wv[:, :, 1].sum(axis=1)
    print(f'{nwg} workgroups; blocks per workgroup: min {blocks.min()} mean {blocks.mean():.1f} max {blocks.max()}')
    print('by XCC: ' + '  '.join(f'{x}: n {np.sum(xcc == x)} end {end[xcc == x].mean():.0f} blk {blocks[xcc == x].mean():.1f}' for x in sorted(set(xcc))))
    print('by SE : ' + '  '.join(f'{x}: n {np.sum(se == x)} end {end[se == x].mean():.0f} blk {blocks[se == x].mean():.1f}' for x in sorted(set(se))))
    print('by CU id: ' + '  '.join(f'{x}: n {np.sum(cu == x)} end {end[cu == x].mean():.0f} blk {blocks[cu == x].mean():.1f}' for x in sorted(set(cu))))
    place = xcc * 1000 + se * 100 + sh * 16 + cu
    per = {}
    for p, e, b in zip(place, end, blocks):
        per.setdefault(int(p), []).append((e, b))
    cnt = np.bincount([len(v) for v in per.values()])
    print(f'distinct (xcc, se, sh, cu) places: {len(per)}; workgroups per place histogram: {list(enumerate(cnt))}')
    for k in sorted(set(len(v) for v in per.values())):
        es = [e for v in per.values() if len(v) == k for e, _ in v]
        bs = [b for v in per.values() if len(v) == k for _, b in v]
        print(f'  places with {k} workgroups: mean end {np.mean(es):.0f} us, blocks per workgroup {np.mean(bs):.1f}')
    wend = (wv[:, :, 0] - t0) / 100.0
    print(f'wave ends: min {wend.min():.0f} p10 {np.percentile(wend, 10):.0f} median {np.median(wend):.0f} p90 {np.percentile(wend, 90):.0f} max {wend.max():.0f}; '
          f'blocks per wave: ' + ' '.join(f'{k}:{v}' for k, v in enumerate(np.bincount(wv[:, :, 1].ravel())) if v))
