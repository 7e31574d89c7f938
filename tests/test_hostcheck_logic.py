"""The kernels' integer logic (2048_amd/csrc/board_ops.hpp, features.hpp), compiled for the HOST by
tests/hostcheck (test-only), against the golden vectors and the oracle.  CPU only — this is how the SWAR
move, spawn, terminal test, RNG and the n-tuple/D4 index code are checked in a container without a GPU.
The real parity tests (through the C-ABI, on the GPU) are in test_gpu_parity.py."""
import ctypes
import importlib
import os
import subprocess

import numpy as np
import pytest

from oracle import ref_batch as rb
from tests.conftest import ROOT

rng_spec = importlib.import_module('2048_amd.rng')
HC_DIR = os.path.join(ROOT, 'tests', 'hostcheck')


@pytest.fixture(scope='module')
def hc():
    so = os.path.join(HC_DIR, 'libhostcheck.so')
    src = os.path.join(HC_DIR, 'hostcheck.cpp')
    subprocess.check_call(['g++', '-O2', '-std=c++17', '-shared', '-fPIC', '-Wno-unknown-pragmas', '-o', so, src])
    return ctypes.CDLL(so)


def ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def hc_move_all(hc, boards):
    boards = np.ascontiguousarray(boards.reshape(-1, 16), np.uint8)
    n = len(boards)
    after = np.zeros((n, 4, 4, 4), np.uint8)
    reward = np.zeros((n, 4), np.int32)
    changed = np.zeros(n, np.uint8)
    hc.hc_move_all(ptr(boards), ctypes.c_int64(n), ptr(after), ptr(reward), ptr(changed))
    return after, reward, changed


def test_every_row_of_the_move_table(hc, golden):
    """All 65 536 rows, in all four directions, against the reference's table."""
    g = golden('move_table.npz')
    keys = np.arange(65536)
    line = np.stack([(keys >> 12) & 15, (keys >> 8) & 15, (keys >> 4) & 15, keys & 15], axis=1).astype(np.uint8)
    boards = np.zeros((65536, 4, 4), np.uint8)
    boards[:, 1, :] = line                                   # the line as row 1 (left / right) ...
    after, reward, changed = hc_move_all(hc, boards)
    assert np.array_equal(after[:, 0, 1, :], g['out'])
    assert np.array_equal(reward[:, 0], g['score'] * g['changed'])
    assert np.array_equal(changed & 1, g['changed'])
    # right = left on the mirrored line
    assert np.array_equal(after[:, 2, 1, ::-1][:, :], hc_move_all(hc, boards[:, :, ::-1])[0][:, 0, 1, :])
    boards = np.zeros((65536, 4, 4), np.uint8)
    boards[:, :, 2] = line                                   # ... and as column 2 (up / down)
    after, reward, changed = hc_move_all(hc, boards)
    assert np.array_equal(after[:, 1, :, 2], g['out'])
    assert np.array_equal(reward[:, 1], g['score'] * g['changed'])
    assert np.array_equal((changed >> 1) & 1, g['changed'])


def test_moves_terminal_spawn_vs_golden(hc, golden):
    g = golden('moves.npz')
    after, reward, changed = hc_move_all(hc, g['boards'])
    assert np.array_equal(after, g['after'])
    assert np.array_equal(reward * g['changed'], g['reward'])    # reference adds score only for changed rows: same sum
    assert np.array_equal(reward, g['reward'])
    bits = np.stack([(changed >> d) & 1 for d in range(4)], axis=1)
    assert np.array_equal(bits, g['changed'])
    n = len(g['boards'])
    b = np.ascontiguousarray(g['boards'].reshape(n, 16))
    over, ne, npairs, top = (np.zeros(n, np.uint8) for _ in range(4))
    hc.hc_terminal(ptr(b), ctypes.c_int64(n), ptr(over), ptr(ne), ptr(npairs), ptr(top))
    assert np.array_equal(over, g['game_over'])
    assert np.array_equal(ne, g['empty_count'])
    assert np.array_equal(npairs, g['adjacent_pair_count'])
    assert np.array_equal(top, b.max(axis=1))
    s = golden('spawn.npz')
    sb = np.ascontiguousarray(s['boards'].reshape(-1, 16)).copy()
    hc.hc_spawn_injected(ptr(sb), ctypes.c_int64(len(sb)), ptr(np.ascontiguousarray(s['r10'])), ptr(np.ascontiguousarray(s['k'])))
    assert np.array_equal(sb.reshape(-1, 4, 4), s['after'])


def test_random_boards_vs_oracle(hc):
    r = np.random.RandomState(5)
    boards = (r.randint(0, 16, (20000, 4, 4)) * (r.rand(20000, 4, 4) < 0.7)).astype(np.uint8)
    after, reward, changed = hc_move_all(hc, boards)
    o_after, o_reward, o_changed = rb.move_all(boards)
    assert np.array_equal(after, o_after)
    assert np.array_equal(reward, o_reward)
    assert np.array_equal(np.stack([(changed >> d) & 1 for d in range(4)], axis=1).astype(bool), o_changed)


@pytest.mark.parametrize('n', [2, 3, 4, 5, 6])
def test_feature_slots_of_all_images(hc, golden, n):
    g = golden('features.npz')
    boards = np.ascontiguousarray(g['boards'].reshape(-1, 16))
    F = g[f'f{n}'].shape[1]
    out = np.zeros((len(boards), 8, F), np.int32)
    assert hc.hc_image_slots(n, ptr(boards), ctypes.c_int64(len(boards)), ptr(out)) == 0
    from oracle import ref_scalar as rs
    offs, total = rs.feature_offsets(n)
    assert np.array_equal(out[:, 0, :] - offs[None, :], g[f'f{n}'])          # identity image == reference f_n
    assert out.min() >= 0 and out.max() < total
    # the 8 images are the 8 the reference's update visits (as a multiset of slot lists)
    imgs = rb.d4_images(g['boards'])
    want = np.stack([rb.slots(n, im) for im in imgs], axis=1)                # [B, 8, F]
    a = np.sort(out.reshape(len(boards), -1), axis=1)
    b = np.sort(want.reshape(len(boards), -1), axis=1)
    assert np.array_equal(a, b)
    # and image by image: ours are indexed g = transpose | mirror_lr << 1 | mirror_ud << 2
    x = g['boards']
    for gi in range(8):
        img = x
        if gi & 2:
            img = img[:, :, ::-1]
        if gi & 4:
            img = img[:, ::-1, :]
        if gi & 1:
            img = np.transpose(img, (0, 2, 1))
        assert np.array_equal(out[:, gi, :], rb.slots(n, img))


@pytest.mark.parametrize('n', [2, 3])
def test_small_orbits_cover_every_add_of_the_reference_update(hc, n):
    """n = 2, 3 (round 3): QAgent.update adds dw at f_i(g.x) for all 8 images g and all features i (r_learning.py:207-214).
    The orbit-reduced update visits, for each orbit representative r, one image per coset of r's stabiliser (SmallOrbits<N>);
    the apply kernel then hands the orbit table to every member feature through a digit permutation and symmetrises over the
    stabiliser.  Checked here by brute force on random boards: (1) the representatives are exactly the first members of the
    D4 orbits of the features; (2) folding the visited images with the stabiliser reproduces the multiset of all 8 images'
    indices of the representative; (3) every feature's indices over the 8 images are a digit permutation of its
    representative's — together: every one of the 8 x F adds of the reference is accounted for, once."""
    F, nd = (24, 2) if n == 2 else (52, 3)
    reps = np.zeros(8, np.int32)
    masks = np.zeros(8, np.uint32)
    count = hc.hc_small_orbits(n, ptr(reps), ptr(masks))
    assert count == (4 if n == 2 else 8)
    rng = np.random.default_rng(11)
    boards = rng.integers(0, 16, size=(96, 16)).astype(np.uint8)
    out = np.zeros((len(boards), 8, F), np.int32)
    assert hc.hc_image_slots(n, ptr(boards), ctypes.c_int64(len(boards)), ptr(out)) == 0
    idx = out - (np.arange(F) * 16 ** nd)[None, None, :]
    digits = np.stack([(idx // 16 ** p) % 16 for p in range(nd)], axis=-1)             # [B, 8, F, nd]

    def match(a, b):                                                                     # perm with a[:, p] == b[:, perm[p]] on every board
        perm, used = [], set()
        for p in range(nd):
            q = next((q for q in range(nd) if q not in used and np.array_equal(a[:, p], b[:, q])), None)
            if q is None:
                return None
            perm.append(q)
            used.add(q)
        return perm

    found, member_of = [], {}
    for i in range(F):
        home = next((r for r in found if any(match(digits[:, 0, i], digits[:, g, r]) is not None for g in range(8))), None)
        if home is None:
            found.append(i)
            home = i
        member_of[i] = home
    assert found == reps[:count].tolist()
    adds = 0
    for o in range(count):
        r, mask = int(reps[o]), int(masks[o])
        stab = [pm for pm in (match(digits[:, 0, r], digits[:, g, r]) for g in range(8)) if pm is not None]
        images = [g for g in range(8) if (mask >> g) & 1]
        assert len(stab) * len(images) == 8
        folded = np.stack([sum(digits[:, g, r, pm[p]] * 16 ** p for p in range(nd)) for g in images for pm in stab], axis=1)
        assert np.array_equal(np.sort(folded, axis=1), np.sort(idx[:, :, r], axis=1))
        members = [i for i in range(F) if member_of[i] == r]
        assert len(members) * len(stab) == 8                                             # orbit-stabiliser
        adds += len(images)
    assert adds == (24 if n == 2 else 52)                                                # instead of 8 x 24 = 192 / 8 x 52 = 416


def test_coset_masks_reproduce_the_eight_images(hc):
    """QAgent.update adds dw at f(g.x) for all 8 images g (r_learning.py:207-214).  The update kernels visit, for the
    representative feature of each symmetry orbit, only the images of COSET_MASK and let the digit permutations of the
    representative's stabiliser supply the rest: as a multiset, {sigma_t(f(g.x)) : g in mask, t in stabiliser} must equal
    {f(g.x) : all 8 g}.  The stabiliser is found here by brute force, as find_orbits does on the device side."""
    reps = np.zeros(8, np.int32)
    masks = np.zeros(8, np.uint32)
    hc.hc_coset_masks(ptr(reps), ptr(masks))
    rng = np.random.default_rng(5)
    boards = rng.integers(0, 14, size=(64, 16)).astype(np.uint8)            # f_6 clamps tiles at 13
    n, F = 6, 33
    out = np.zeros((len(boards), 8, F), np.int32)
    assert hc.hc_image_slots(n, ptr(boards), ctypes.c_int64(len(boards)), ptr(out)) == 0
    from oracle import ref_scalar as rs
    offs, _ = rs.feature_offsets(n)
    adds = 0
    for rep, mask in zip(reps, masks):
        radix, nd = (16, 4) if rep < 17 else (16, 5) if rep < 21 else (14, 6)
        idx = out[:, :, rep] - offs[rep]                                     # [B, 8]
        digits = np.stack([(idx // radix ** p) % radix for p in range(nd)], axis=-1)      # [B, 8, nd]
        stab = []
        for g in range(8):                                                   # f(x) = perm(f(g.x)) on every board?
            perm, used = [], set()
            for p in range(nd):
                q = next((q for q in range(nd) if q not in used and np.array_equal(digits[:, 0, p], digits[:, g, q])), None)
                if q is None:
                    break
                perm.append(q)
                used.add(q)
            if len(perm) == nd:
                stab.append(perm)
        images = [g for g in range(8) if (mask >> g) & 1]
        assert len(stab) * len(images) == 8, (rep, stab, images)
        folded = np.stack([sum(digits[:, g, perm[p]] * radix ** p for p in range(nd)) for g in images for perm in stab], axis=1)
        assert np.array_equal(np.sort(folded, axis=1), np.sort(idx, axis=1)), rep
        adds += len(images)
    assert adds == 4 + 4 + 4 + 4 + 1 + 4 + 8 + 4


def test_rng_stream_and_new_games(hc):
    seed, lane0, count, nd = 2048, 1000, 64, 16
    draws = np.zeros((count, nd), np.uint64)
    state = np.zeros((count, 2), np.uint64)
    hc.hc_rng_stream(ctypes.c_uint64(seed), ctypes.c_uint64(lane0), ctypes.c_int64(count), nd, ptr(draws), ptr(state))
    st = rng_spec.seed_lanes(seed, lane0, count)
    for j in range(nd):
        assert np.array_equal(rng_spec.next_u64_np(st), draws[:, j])
    assert np.array_equal(st, state)
    lane = rng_spec.LaneRng(seed, lane0 + 5)
    assert [lane.next() for _ in range(nd)] == [int(v) for v in draws[5]]
    boards = np.zeros((count, 16), np.uint8)
    hc.hc_new_games(ctypes.c_uint64(seed), ctypes.c_uint64(lane0), ctypes.c_int64(count), ptr(boards))
    st = rng_spec.seed_lanes(seed, lane0, count)
    want = np.zeros((count, 4, 4), np.uint8)
    for _ in range(2):
        r10, k = rng_spec.spawn_draw_np(rng_spec.next_u64_np(st), rb.empty_count(want))
        want, _, _ = rb.spawn_injected(want, r10, k)
    assert np.array_equal(boards.reshape(-1, 4, 4), want)


def test_table_places_are_bijections_and_memory_slots_agree(hc, golden):
    """The table is stored in another order than it is indexed (features.hpp: table_place / hex_place).  Both maps must
    be bijections of their block, and memory_slots<n> — what the kernels address the table with — must be the place of
    feature_slots<n>, for every n (n = 2, 3: unchanged)."""
    from oracle import ref_scalar as rs
    for block in (0, 5, 16, 17, 40, 80):                     # 65 536-entry blocks of the four- and five-cell tables
        slots = (np.arange(65536, dtype=np.uint32) + np.uint32(block * 65536))
        out = np.zeros_like(slots)
        hc.hc_table_place(ptr(slots), ctypes.c_int64(len(slots)), ptr(out))
        assert np.array_equal(np.sort(out), slots)
        assert not np.array_equal(out, slots)
    hexn = 14 ** 6
    out = np.zeros(hexn, np.uint32)
    hc.hc_hex_place(ctypes.c_uint32(0), ctypes.c_int64(hexn), ptr(out))
    assert np.array_equal(np.sort(out), np.arange(hexn, dtype=np.uint32))
    k = np.arange(hexn, dtype=np.int64)                       # the stated formula: 64 * sum (d_p >> 1) 7^p + sum (d_p & 1) 2^p
    want = np.zeros(hexn, np.int64)
    for p in range(6):
        d = (k // 14 ** p) % 14
        want += 64 * (d >> 1) * 7 ** p + ((d & 1) << p)
    assert np.array_equal(out.astype(np.int64), want)
    g = golden('features.npz')
    boards = np.ascontiguousarray(g['boards'].reshape(-1, 16))
    extra = np.random.RandomState(3).randint(0, 16, size=(4096, 16)).astype(np.uint8)     # tiles 14, 15 too (the base-14 clamp)
    boards = np.ascontiguousarray(np.concatenate([boards, extra]))
    for n in (2, 3, 4, 5, 6):
        offs, total = rs.feature_offsets(n)
        F = len(offs)
        logical = np.zeros((len(boards), 8, F), np.int32)
        assert hc.hc_image_slots(n, ptr(boards), ctypes.c_int64(len(boards)), ptr(logical)) == 0
        logical = np.ascontiguousarray(logical[:, 0, :]).astype(np.uint32)
        mem = np.zeros((len(boards), F), np.int32)
        assert hc.hc_memory_slots(n, ptr(boards), ctypes.c_int64(len(boards)), ptr(mem)) == 0
        if n < 4:
            assert np.array_equal(mem.astype(np.uint32), logical)
            continue
        placed = np.zeros(logical.size, np.uint32)
        flat = np.ascontiguousarray(logical.reshape(-1))
        hc.hc_table_place(ptr(flat), ctypes.c_int64(flat.size), ptr(placed))
        assert np.array_equal(mem.astype(np.uint32).reshape(-1), placed)
        assert placed.max() < total
        # a feature's entries stay inside its own table
        lo = np.asarray(offs, np.int64)[None, :]
        hi = np.append(np.asarray(offs, np.int64)[1:], total)[None, :]
        m = mem.astype(np.int64)
        assert (m >= lo).all() and (m < hi).all()


def test_cross_order_is_a_bijection_with_the_cells_high_bits_on_top(hc):
    """The cross orbit's accumulation table is indexed by cross_order (features.hpp): a bijection of the 20-bit index, its
    inverse cross_unorder, and bit 5 b + j of the result = bit b of cell j (so the top five bits are bit 3 of the cells)."""
    n = 1 << 20
    fwd, back = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
    hc.hc_cross_order(ctypes.c_int64(n), ptr(fwd), ptr(back))
    k = np.arange(n, dtype=np.uint32)
    assert np.array_equal(np.sort(fwd), k)
    assert np.array_equal(back[fwd], k) and np.array_equal(fwd[back], k)
    cells = [(k >> (4 * j)) & 15 for j in range(5)]
    want = np.zeros(n, np.uint32)
    for b in range(4):
        for j in range(5):
            want |= ((cells[j] >> b) & 1).astype(np.uint32) << np.uint32(5 * b + j)
    assert np.array_equal(fwd, want)


def test_quad_place_order_of_the_four_cell_orbit_tables(hc):
    """The four-cell orbits' accumulation tables (features.hpp, quad_place): the 11^4 indices with every cell <= 10 first, as
    base-11 numbers inside the first 16 384 slots, then the 16-bit index space in index order of which only indices with a cell
    >= 11 are used; quad_unplace is the inverse and reports the holes; the two-indices-per-word forms the owner kernel uses
    agree with the scalar ones on every half."""
    dsize = 16384 + 65536
    place, hot, unplace = np.zeros(65536, np.uint32), np.zeros(65536, np.uint8), np.zeros(dsize, np.int32)
    rng = np.random.default_rng(5)
    words = np.concatenate([rng.integers(0, 1 << 32, 200000, dtype=np.uint64).astype(np.uint32),
                            (np.arange(65536, dtype=np.uint32) << 16) | np.arange(65536, dtype=np.uint32)[::-1]])
    big, b11 = np.zeros(len(words), np.uint32), np.zeros(len(words), np.uint32)
    assert hc.hc_quad_place(ptr(place), ptr(hot), ptr(unplace), ctypes.c_int64(len(words)), ptr(words), ptr(big), ptr(b11)) == dsize
    k = np.arange(65536, dtype=np.int64)
    cells = np.stack([(k >> s) & 15 for s in (12, 8, 4, 0)], axis=1)
    want_hot = (cells <= 10).all(axis=1)
    assert np.array_equal(hot.astype(bool), want_hot)
    base11 = ((cells[:, 0] * 11 + cells[:, 1]) * 11 + cells[:, 2]) * 11 + cells[:, 3]
    assert np.array_equal(place.astype(np.int64), np.where(want_hot, base11, 16384 + k))
    assert len(np.unique(place)) == 65536 and place[want_hot].max() == 11 ** 4 - 1
    used = np.zeros(dsize, bool)
    used[place] = True
    assert np.array_equal(unplace >= 0, used)                               # every other slot is a hole
    assert np.array_equal(unplace[place], k)
    for half, w in ((0, words & 0xFFFF), (1, words >> 16)):
        w = w.astype(np.int64)
        bh = (big >> (16 * half)) & 0xFFFF
        assert np.array_equal(bh == 0, want_hot[w])
        ok = want_hot[w]
        assert np.array_equal(((b11 >> (16 * half)) & 0xFFFF)[ok].astype(np.int64), base11[w][ok])
