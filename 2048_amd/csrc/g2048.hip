// g2048.hip — HIP kernels (gfx950 / MI355X) and the C ABI of include/g2048.h.
//
// One lane = one board = one thread; a wavefront holds 64 boards, a workgroup 256.  Lane state is
// struct-of-arrays in HBM so that every load/store is a fully coalesced wave access:
//   boards  uint4[B]      16 B   log2 tiles, row-major (game_logic.py:62 narrowed to u8)
//   scores  int32[B]       4 B   Game.score
//   rng     u64[B][2]     16 B   xoroshiro128++ state (spec: 2048_amd/rng.py)
//   prev    uint4[2][B]   16 B   `state` of QAgent.episode (r_learning.py:226,245) in packed form (4 row + 4 column 16-bit
//                                indices, features.hpp), double-buffered so that the step's two update records need no
//                                extra copy (see k_td_play)
//   label   float[B]       4 B   `old_label`
//   flags   u8[B]                HAS_PREV / DONE
//   lane_id u32[B]               which lane of the context sits at this position (the lanes are re-ordered by board pattern
//                                every few steps, LaneSort; the ABI always shows them in lane order)
//   weights float[slots]         flat n-tuple table, feature-major, weight_signature group order
// One TD step = k_td_play (all lanes choose and move with the same table) -> k_td_update_owner (the step's records are
// summed per symmetry orbit by workgroups that own table slices in LDS; n = 6: + k_hex_* for the f_6 orbits) ->
// k_apply_orbits (orbit sums -> member tables).
// There is no dense contraction anywhere on this path: no MFMA.  The kernels are integer SWAR + random 4-byte
// gathers + fp32 atomic adds; what bounds them is the memory system, not the VALU.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>      // types only: the library is bound with dlopen (g2048_comm_init)
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/g2048.h"
#include "features.hpp"
#include "lane_state.hpp"
#include "lookahead.hpp"
#include "play.hpp"

using namespace g2048;

namespace {

// ------------------------------------------------------------------------------------------------ lane state

__global__ __launch_bounds__(WG) void k_seed(ulonglong2* rng, uint32_t B, uint64_t seed, uint64_t lane0) {
    uint32_t i = blockIdx.x * WG + threadIdx.x;
    if (i < B) st_rng(rng, i, seed_lane(seed, lane0 + i));
}

// Game.__init__ (game_logic.py:55-66) in every lane
__global__ __launch_bounds__(WG) void k_new_games(uint4* boards, int32_t* scores, ulonglong2* rng, float* label, uint8_t* flags,
                                                  uint32_t B) {
    uint32_t i = blockIdx.x * WG + threadIdx.x;
    if (i >= B) return;
    Rng g = ld_rng(rng, i);
    st_board(boards, i, new_game(g));
    st_rng(rng, i, g);
    scores[i] = 0;
    label[i] = 0.0f;
    flags[i] = 0;
}

// ------------------------------------------------------------------------------------------------ environment

// Game.pre_move x 4 (game_logic.py:136-142)
__global__ __launch_bounds__(WG) void k_move_all(const uint4* boards, uint32_t B, uint4* after, int4* reward, uint8_t* changed) {
    uint32_t i = blockIdx.x * WG + threadIdx.x;
    if (i >= B) return;
    Moves4 mv = all_moves(ld_board(boards, i));
    st_board(after, (size_t)i * 4 + 0, mv.m0.after);
    st_board(after, (size_t)i * 4 + 1, mv.m1.after);
    st_board(after, (size_t)i * 4 + 2, mv.m2.after);
    st_board(after, (size_t)i * 4 + 3, mv.m3.after);
    reward[i] = make_int4((int32_t)merged_score(mv.m0.ma, mv.m0.mb), (int32_t)merged_score(mv.m1.ma, mv.m1.mb),
                          (int32_t)merged_score(mv.m2.ma, mv.m2.mb), (int32_t)merged_score(mv.m3.ma, mv.m3.mb));
    changed[i] = changed_mask(mv);
}

// Game.make_move (game_logic.py:144-148)
__global__ __launch_bounds__(WG) void k_apply_moves(uint4* boards, int32_t* scores, uint32_t B, const uint8_t* dirs, uint8_t* moved) {
    uint32_t i = blockIdx.x * WG + threadIdx.x;
    if (i >= B) return;
    Moves4 mv = all_moves(ld_board(boards, i));
    uint32_t d = dirs[i] & 3u;
    Moved c = pick(mv, d);
    st_board(boards, i, c.after);
    scores[i] += (int32_t)merged_score(c.ma, c.mb);
    if (moved) moved[i] = c.changed;
}

__global__ __launch_bounds__(WG) void k_terminal(const uint4* boards, uint32_t B, uint8_t* over, uint8_t* n_empty, uint8_t* n_pairs) {
    uint32_t i = blockIdx.x * WG + threadIdx.x;
    if (i >= B) return;
    Board b = ld_board(boards, i);
    over[i] = game_over(b);
    n_empty[i] = (uint8_t)empty_count(b);
    n_pairs[i] = (uint8_t)adjacent_pairs(b);
}

// Game.new_tile (game_logic.py:112-121): injected draws, or the lane's own stream with the draws exported
__global__ __launch_bounds__(WG) void k_spawn(uint4* boards, ulonglong2* rng, uint32_t B, const uint8_t* in_r10, const uint8_t* in_k,
                                              uint8_t* out_r10, uint8_t* out_k) {
    uint32_t i = blockIdx.x * WG + threadIdx.x;
    if (i >= B) return;
    Board b = ld_board(boards, i);
    uint32_t e = empty_bits(b);
    uint32_t r10 = 255, k = 255;
    if (e) {
        if (in_r10) {
            r10 = in_r10[i];
            k = in_k[i];
            if (k >= popcount32(e)) k = popcount32(e) - 1;      // never index past the empty list
        } else {
            Rng g = ld_rng(rng, i);
            spawn_draw(next_u64(g), popcount32(e), r10, k);
            st_rng(rng, i, g);
        }
        place_tile(b, r10, k, e);
        st_board(boards, i, b);
    }
    if (out_r10) out_r10[i] = (uint8_t)r10;
    if (out_k) out_k[i] = (uint8_t)k;
}

// BASELINE config 2: nsteps of {uniformly random valid direction, move, spawn, terminal check / auto-reset}
// with the lane state held in registers for the whole launch.
__global__ __launch_bounds__(WG) void k_step_random(uint4* boards, int32_t* scores, ulonglong2* rng, uint8_t* flags, uint32_t B,
                                                    uint32_t nsteps, int auto_reset, Stats* stats) {
    __shared__ WgStats ws;
    wg_stats_init(&ws);
    uint32_t i = blockIdx.x * WG + threadIdx.x;
    const bool in = i < B;
    Board b = {{0, 0, 0, 0}};
    Rng g = {1, 0};
    int32_t score = 0;
    uint8_t fl = DONE;
    if (in) {
        b = ld_board(boards, i);
        g = ld_rng(rng, i);
        score = scores[i];
        fl = flags[i];
    }
    uint32_t my_moves = 0, my_dirs = 0;
    for (uint32_t s = 0; s < nsteps; ++s) {
        if (fl & DONE) continue;
        Moves4 mv = all_moves(b);
        uint32_t mask = changed_mask(mv);
        bool over;
        if (mask) {
            uint32_t j = pick_draw(next_u64(g), popcount32(mask));
            uint32_t d = kth_set_bit(mask, j);
            Moved c = pick(mv, d);
            b = c.after;
            score += (int32_t)merged_score(c.ma, c.mb);
            ++my_moves;
            my_dirs += popcount32(mask);
            spawn(b, g);
            over = game_over(b) || max_tile(b) >= 16u;
        } else {
            over = true;                                        // a dead board was loaded
        }
        if (over) {
            count_finished(&ws, b, score, max_tile(b) >= 16u);
            if (auto_reset) {
                b = new_game(g);
                score = 0;
            } else {
                fl |= DONE;
            }
        }
    }
    count_moves(&ws, my_moves, my_dirs);
    wg_stats_flush(&ws, stats);
    if (in) {
        st_board(boards, i, b);
        st_rng(rng, i, g);
        scores[i] = score;
        flags[i] = fl;
    }
}

// ------------------------------------------------------------------------------------------------ features / value

template <int N>
__global__ __launch_bounds__(WG) void k_features(const uint4* boards, uint32_t B, int32_t* out) {
    uint32_t i = blockIdx.x * WG + threadIdx.x;
    if (i >= B) return;
    constexpr int F = Shape<N>::F;
    uint32_t s[F];
    feature_slots<N>(pack_board(ld_board(boards, i)), s);
#pragma unroll
    for (int f = 0; f < F; ++f) out[(size_t)i * F + f] = (int32_t)(s[f] - feature_offset(N, f));
}

template <int N>
__global__ __launch_bounds__(WG) void k_evaluate(const uint4* boards, uint32_t B, const float* __restrict__ w, float* value) {
    uint32_t i = blockIdx.x * WG + threadIdx.x;
    if (i >= B) return;
    value[i] = value_of<N>(w, ld_board(boards, i));
}

// ------------------------------------------------------------------------------------------------ learning

// QAgent.update (r_learning.py:207-214), one of the 8 images: += dw at every feature slot
// `dacc` (may be null): the epoch's accumulated weight delta of the multi-GPU scheme — every add to the table is
// mirrored there (g2048_delta_begin)
template <int N>
__device__ __forceinline__ void scatter_image(float* w, float* dacc, const Packed& state, uint32_t g, float dw) {
    constexpr int F = Shape<N>::F;
    uint32_t s[F];
    memory_slots<N>(d4_image(state, g), s);
#pragma unroll
    for (int f = 0; f < F; ++f) __hip_atomic_fetch_add(&w[s[f]], dw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (dacc) {
#pragma unroll
        for (int f = 0; f < F; ++f) __hip_atomic_fetch_add(&dacc[s[f]], dw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// thread t handles image (t & 7) of record (t >> 3)
template <int N>
__global__ __launch_bounds__(WG) void k_update_records(float* w, float* dacc, const uint4* states, const float* dw, uint32_t count) {
    uint32_t t = blockIdx.x * WG + threadIdx.x;
    uint32_t rec = t >> 3;
    if (rec >= count) return;
    scatter_image<N>(w, dacc, pack_board(ld_board(states, rec)), t & 7u, dw[rec]);
}


__global__ __launch_bounds__(WG) void k_log_init(GameLog lg, const uint4* boards, const uint8_t* flags) {
    uint32_t i = blockIdx.x * WG + threadIdx.x;
    if (i >= lg.lanes) return;
    uint32_t* m = lg.meta + 8 * i;
#pragma unroll
    for (int j = 0; j < 8; ++j) m[j] = 0;
    m[7] = (flags[i] & HAS_PREV) ? LOG_PARTIAL0 : 0u;       // a game already under way cannot be replayed from its start
    lg.start[2 * i] = boards[i];
}


// Step part 2, global-atomics form — QAgent.update for every record: thread t adds image (t & 7) of record (t >> 3);
// records B .. B + qcount - 1 are the terminal queue.
template <int N>
__global__ __launch_bounds__(WG) void k_td_update(float* w, float* dacc, TdRecs recs, uint32_t B) {
    uint32_t t = blockIdx.x * WG + threadIdx.x;
    uint32_t r = t >> 3, g = t & 7u;
    if (r < B) {
        float dw = recs.dw1[r];
        if (dw != 0.0f) scatter_image<N>(w, dacc, ld_packed(recs.state1, r), g, dw);
    } else if (r - B < *recs.qcount) {
        scatter_image<N>(w, dacc, ld_packed(recs.qstate, r - B), g, recs.qdw[r - B]);
    }
}

// ------------------------------------------------------------------------------------------------ LDS-owner update
// Random fp32 atomics to HBM-side memory run at ~21 G adds/s on MI355X and collapse under the skew of real boards
// (a few hot slots: 7.8 G/s measured, tools/atomic_bench.hip); k_td_update spends 22 ms per step of 2^20 lanes on
// them.  The chip has 256 x 160 KiB = 40 MiB of LDS, more than the whole n <= 5 table (21 MB): so instead a
// workgroup OWNS a 128 KiB slice of the table in LDS, streams the step's records (n >= 4: its orbit's precomputed indices + dw, 12 B; else packed state + dw, 20 B; L2/MALL
// resident), computes for each record and each of the 8 images only the slots of ITS feature(s), accumulates the
// hits with LDS atomics, and finally adds the slice to the table in HBM with coalesced accesses.  With one
// workgroup per slice the flush is a plain read-modify-write (no global atomics at all); when the records are split
// over `nparts` workgroups the flush uses contiguous atomics.
struct Slice {
    uint32_t variant;       // which feature(s) this workgroup encodes (see OwnVariants)
    uint32_t tlo, size;     // table slots [tlo, tlo + size) of that feature are held in LDS, size <= OWN_SLOTS
    uint32_t dlo;           // where the slice is added at flush time: index into the orbit table D (n >= 4) or into w (n = 2, 3)
    uint32_t part, nparts;  // records [count * part / nparts, count * (part + 1) / nparts)
    uint32_t chunk;         // hit counter of this slice (load statistics for the planner)
    // Chunks of the same orbit table that nobody holds in LDS (they see too few adds to be worth a record scan) are
    // added straight into D with global atomics by the orbit's busiest chunk: bit k of fb_mask = chunk k of the orbit
    // is this workgroup's duty.  orb_tlo / orb_dlo / chunk0: first table slot, first D index, first hit counter of the orbit.
    uint32_t orb_tlo, orb_dlo, chunk0;
    uint64_t fb_mask;
    uint32_t cshift;        // log2 of the slots per chunk of this orbit table (OWN_SLOTS, or OWN_SLOTS / 2 for the fixed-point orbits)
    uint32_t fixed;         // 1: this workgroup sums in 64-bit fixed point (own_fixed, or every variant under the one-pass mean rule)
    uint32_t xcd;           // 1: XCD-resident plan — part = x * (nparts / 8) + j scans the record blocks b = x + 8 * (j + (nparts / 8) * t)
    uint32_t quad_plain;    // 1: the four-cell orbits' tables are used in index order only (no index is "hot": QuadOrder below)
};

// Symmetry orbits (n >= 4).  QAgent.update adds dw at f_i(g.x) for all 8 images g (r_learning.py:207-214).  Features
// that the dihedral group maps onto each other (the 4 outer lines; the 4 inner lines; corner / edge / centre squares;
// the 4 crosses) therefore receive the SAME multiset of adds up to a fixed permutation of the index nibbles:
//     f_i(x) = perm_i(f_rep(h_i . x))  for all x   =>   delta_table_i[perm_i(k)] = D_rep[k].
// So the owner kernel accumulates one table D per orbit (6 orbits instead of 21 features: 3.5x fewer index
// computations, LDS adds and record scans) and k_apply_orbits adds D into every member table through its permutation.
constexpr int MAX_ORBITS = 8, MAX_MEMBERS = 8;
struct OrbitInfo {
    uint32_t base, size;                // D[base .. base + size)
    uint32_t nmem;
    uint32_t off[MAX_MEMBERS];          // first table slot of each member feature
    uint32_t perm[MAX_MEMBERS];         // 3 bits per output digit p: the input digit it takes (out digit p = in digit src[p])
    uint32_t digits, radix;             // 4 or 5 digits in base 16 (nibbles), or 6 digits in base 14 (the f_6 features)
    // Stabiliser of the representative: the images g with f_rep(g.x) = sigma_g(f_rep(x)) (the feature's cells map onto
    // themselves; e.g. the up-down mirror reverses the nibbles of a column).  The owner kernel then visits only one image
    // per coset (COSET_MASK) and accumulates E; the orbit table is D[k] = sum over the stabiliser of E[sigma(k)].
    uint32_t nstab;
    uint32_t sperm[8];                  // digit permutations of the stabiliser (sperm[0] = identity), encoded like perm
};
struct OrbitTable {
    uint32_t count, total;
    OrbitInfo o[MAX_ORBITS];
};

G2048_HD uint32_t permute_digits(uint32_t k, uint32_t perm, uint32_t digits, uint32_t radix) {
    if (radix == 16u) {
        uint32_t out = 0;
        for (uint32_t p = 0; p < digits; ++p) out |= ((k >> (4u * ((perm >> (3u * p)) & 7u))) & 15u) << (4u * p);
        return out;
    }
    uint32_t d[6], out = 0, mul = 1;
    for (uint32_t p = 0; p < 6u; ++p) {
        d[p] = k % 14u;
        k /= 14u;
    }
    for (uint32_t p = 0; p < 6u; ++p) {
        const uint32_t src = (perm >> (3u * p)) & 7u;
        uint32_t v = d[0];                      // select without dynamic indexing
        v = src == 1u ? d[1] : v; v = src == 2u ? d[2] : v; v = src == 3u ? d[3] : v; v = src == 4u ? d[4] : v; v = src == 5u ? d[5] : v;
        out += v * mul;
        mul *= 14u;
    }
    return out;
}

// permute_digits for a base-14 index, delivered at its place in memory (hex_place of the result, without a second digit extraction)
G2048_HD uint32_t permute_hex_placed(uint32_t k, uint32_t perm) {
    uint32_t d[6], hi = 0, lo = 0, m7 = 1;
    for (uint32_t p = 0; p < 6u; ++p) {
        d[p] = k % 14u;
        k /= 14u;
    }
    for (uint32_t p = 0; p < 6u; ++p) {
        const uint32_t src = (perm >> (3u * p)) & 7u;
        uint32_t v = d[0];
        v = src == 1u ? d[1] : v; v = src == 2u ? d[2] : v; v = src == 3u ? d[3] : v; v = src == 4u ? d[4] : v; v = src == 5u ? d[5] : v;
        hi += (v >> 1) * m7;
        lo |= (v & 1u) << p;
        m7 *= 7u;
    }
    return 64u * hi + lo;
}

#ifndef G2048_OWN_U
#define G2048_OWN_U 4       // records per thread in flight in the owner kernel's scan (n >= 4)
#endif
constexpr int OWN_WG = 1024;
constexpr uint32_t OWN_SLOTS = 32768;      // 128 KiB of the CU's 160 KiB LDS
template <int FC> struct OwnUnroll { static constexpr int U = FC == 1 ? 4 : 1; };   // records in flight per thread (loads issued together)

// variant v of table N covers features [f0(v), f0(v) + fc(v)); n >= 4: the orbit representatives ORBIT_REPS
template <int N> struct OwnVariants { static constexpr int COUNT = N == 4 ? 5 : 6; static constexpr int f0(int v) { return ORBIT_REPS[v]; } static constexpr int fc(int) { return 1; } };
// n = 2, 3: a variant is a chunk of PER_CHUNK whole orbit tables (n = 2: all four, 1 024 slots; n = 3: 4 x 4 096 fixed-point slots = 128 KiB, two chunks)
template <> struct OwnVariants<2> { static constexpr int COUNT = 1; static constexpr int f0(int) { return 0; } static constexpr int fc(int) { return 24; } };
template <> struct OwnVariants<3> { static constexpr int COUNT = 2; static constexpr int f0(int) { return 0; } static constexpr int fc(int) { return 52; } };

#ifndef G2048_FIXED_VARIANTS
#define G2048_FIXED_VARIANTS 6      // (5 = the cross orbit in fp32: half as many, twice as large chunks, but ds_add_f32 is 12x slower than ds_add_u64)
#endif
// which variants sum in 64-bit fixed point: the five four-cell orbits of n >= 4, and every feature group of n = 2, 3
// (n = 3: 2.25 -> 0.36 ms per update, n = 2: 1.07 -> 0.24 ms)
constexpr bool own_fixed(int n, int variant) { return n < 4 || variant < G2048_FIXED_VARIANTS; }
constexpr uint32_t FIXED_SLOTS = OWN_SLOTS / 2;

__device__ __forceinline__ Packed unpack4(const uint4& v) {
    Packed q;
    q.R[0] = v.x & 0xFFFFu; q.R[1] = v.x >> 16; q.R[2] = v.y & 0xFFFFu; q.R[3] = v.y >> 16;
    q.C[0] = v.z & 0xFFFFu; q.C[1] = v.z >> 16; q.C[2] = v.w & 0xFFFFu; q.C[3] = v.w >> 16;
    return q;
}

// dw * 2^S rounded to the nearest (even) integer, as a 64-bit two's complement number: adding 1.5 * 2^52 leaves exactly
// that integer in the low mantissa bits of the double (|dw * 2^S| < 2^39 here).  Four instructions; the float -> int64
// conversion of the compiler's runtime is fourteen.
__device__ __forceinline__ long long to_fixed(float dw, double scale) {
    constexpr double MAGIC = 6755399441055744.0;
    return __double_as_longlong(fma((double)dw, scale, MAGIC)) - __double_as_longlong(MAGIC);
}

// FIXED: the four-cell orbits (n >= 4) take 17 of a record's 21 adds (40 of 48 before the coset reduction), and `ds_add_f32` manages 0.33 lane-adds per
// cycle per CU against 4.1 for `ds_add_u64` (profiles/r01_lds_atomic_microbench.txt).  Their workgroups therefore sum
// in 64-bit fixed point: dw * 2^S with S chosen from the step's largest |dw| so that 2^24 adds cannot overflow and a
// dw 2^-14 times smaller than the largest is still exact; the flush converts back.  Half as many slots fit in LDS
// (chunks of 16 384), the sums no longer depend on the order of the adds.
// One-pass mean rule (cbits > 0): the low `cbits` bits of a fixed-point slot count the adds, the rest is the sum — every
// add is (dw * 2^S << cbits) + 1, the flush splits the word again (k_td_update_owner).  cbits = 0: sums only.
__device__ __forceinline__ unsigned long long packed_add(float dw, double scale, uint32_t cbits) {
    const long long fixed = to_fixed(dw, scale);
    return ((unsigned long long)fixed << cbits) + (cbits ? 1ull : 0ull);
}

// A fallback add — to a chunk of the orbit that no workgroup holds — goes straight to D.  (A small {D slot, packed sum} cache in
// the 32 KB of LDS the accumulators leave free was tried for them: a compare-and-swap round trip inside divergent code per add;
// it cut the cost of a 3 % fallback share from 70 to 27 us per launch and slowed a trained agent's step, whose fallback adds are
// spread thin, by 2 - 4 %: profiles/r04_experiments.txt item 11.)
__device__ __forceinline__ void fb_add(float* D, float* Dc, uint32_t dslot, float dw) {
    __hip_atomic_fetch_add(&D[dslot], dw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (Dc) __hip_atomic_fetch_add(&Dc[dslot], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int N, int F0, int FC, bool FB, bool FIXED, uint32_t IMAGES, bool QPLAIN>
__device__ __forceinline__ void own_accum(const Packed& p, float dw, bool valid, float* acc, const Slice& sl, uint32_t& nhit, float* D,
                                          float* Dc, uint32_t* fb_hits, float scale, uint32_t cbits) {
    constexpr int F = Shape<N>::F;
    unsigned long long fixed = 0;
    if (FIXED) fixed = packed_add(dw, (double)scale, cbits);
#pragma unroll
    for (uint32_t g = 0; g < 8; ++g) {
        if (!((IMAGES >> g) & 1u)) continue;
        uint32_t s[F];
        feature_slots<N>(d4_image(p, g), s);         // g is a constant after unrolling; unused slots are dead code
#pragma unroll
        for (int f = F0; f < F0 + FC; ++f) {
            const bool cross = N >= 5 && f >= 17 && f < 21;                 // the cross orbit's table is in cross_order,
            const uint32_t rel16 = s[f] - sl.orb_tlo;                       // a four-cell orbit's in quad_place order (features.hpp)
            const uint32_t orel = cross ? cross_order(rel16) : f < 17 ? (QPLAIN ? QUAD_HOT + rel16 : quad_place(rel16)) : rel16;
            const uint32_t local = orel - (sl.tlo - sl.orb_tlo);
            const bool hit = valid && local < sl.size;
            if (hit) {
                if (FIXED)
                    atomicAdd(reinterpret_cast<unsigned long long*>(acc) + local, fixed);
                else
                    atomicAdd(&acc[local], dw);
            }
            nhit += hit ? 1u : 0u;
            if (FB && valid && !hit) {
                const uint32_t rel = orel, ch = rel >> sl.cshift;
                if ((sl.fb_mask >> ch) & 1u) {
                    fb_add(D, Dc, sl.orb_dlo + rel, dw);
                    atomicAdd(&fb_hits[ch], 1u);            // LDS counter, flushed once per workgroup
                }
            }
        }
    }
}

// n = 2, 3: chunk V holds the orbit tables [V * PER_CHUNK, (V + 1) * PER_CHUNK) whole, so every add of those orbits is a hit;
// one image per coset of each representative's stabiliser (SmallOrbits<N>::mask), always in fixed point.
template <int N, int V>
__device__ __forceinline__ void own_accum_small(const Packed& p, float dw, bool valid, float* acc, uint32_t& nhit, float scale, uint32_t cbits) {
    using SO = SmallOrbits<N>;
    constexpr int F = Shape<N>::F, O0 = V * SO::PER_CHUNK;
    const unsigned long long fixed = packed_add(dw, (double)scale, cbits);
    unsigned long long* const slots = reinterpret_cast<unsigned long long*>(acc);
#pragma unroll
    for (uint32_t g = 0; g < 8; ++g) {
        uint32_t s[F];
        feature_slots<N>(d4_image(p, g), s);            // g is a constant after unrolling; what no orbit of the chunk visits is dead code
#pragma unroll
        for (int j = 0; j < SO::PER_CHUNK; ++j)
            if ((SO::mask(O0 + j) >> g) & 1u) {
                const uint32_t local = (uint32_t)j * SO::SIZE + (s[SO::rep(O0 + j)] - feature_offset(N, SO::rep(O0 + j)));
                if (valid) atomicAdd(slots + local, fixed);
                nhit += valid ? 1u : 0u;
            }
    }
}

// the same accumulation from precomputed orbit indices (k_td_play's OrbitIdx records); idx are relative to the orbit table
// (BIAS: what a fallback add puts on top of the index to get its place — QUAD_HOT for the four-cell orbits in plain order, whose
// records hold plain indices and whose `lo_rel` the caller has lowered by as much)
template <int NI, bool FB, bool FIXED, uint32_t BIAS = 0u>
__device__ __forceinline__ void own_accum_idx(const uint32_t (&idx)[NI], float dw, bool valid, float* acc, const Slice& sl, uint32_t lo_rel,
                                              uint32_t& nhit_lane, float* D, float* Dc, uint32_t* fb_hits, double scale, uint32_t cbits) {
    // A scan is bound by instruction issue (~0.46 ns per record and workgroup whether the record is 12 or 20 bytes,
    // profiles/r02_owner_experiments.txt), and most workgroups hit with few of their records: the hits are counted per lane
    // (one add with the compare's mask as carry) and the fixed-point form of dw is only made for a record that hits.
    bool hit[NI];
    bool any = false;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        hit[j] = valid && idx[j] - lo_rel < sl.size;
        any |= hit[j];
        nhit_lane += hit[j] ? 1u : 0u;
    }
    if (any) {
        unsigned long long fixed = 0;
        if (FIXED) fixed = packed_add(dw, scale, cbits);
#pragma unroll
        for (int j = 0; j < NI; ++j)
            if (hit[j]) {
                const uint32_t local = idx[j] - lo_rel;
                if (FIXED)
                    atomicAdd(reinterpret_cast<unsigned long long*>(acc) + local, fixed);
                else
                    atomicAdd(&acc[local], dw);
            }
    }
    if (FB && valid) {
#pragma unroll
        for (int j = 0; j < NI; ++j)
            if (!hit[j]) {
                const uint32_t rel = idx[j] + BIAS, ch = rel >> sl.cshift;
                if ((sl.fb_mask >> ch) & 1u) {
                    fb_add(D, Dc, sl.orb_dlo + rel, dw);
                    atomicAdd(&fb_hits[ch], 1u);            // LDS counter, flushed once per workgroup
                }
            }
    }
}

// NI indices (4: words x, y; 1: the low half of x) of one record into the chunk of this workgroup: HOT = chunk 0 (all cells <= 10),
// otherwise the chunk of the indices [lo16, lo16 + 16 384) that have a cell >= 11.
template <int NI, bool FB, bool HOT>
__device__ __forceinline__ void own_accum_quad(uint32_t x, uint32_t y, float dw, bool valid, float* acc, const Slice& sl, uint32_t lo16,
                                               uint32_t& nhit_lane, float* D, float* Dc, uint32_t* fb_hits, double scale, uint32_t cbits) {
    const uint32_t w[2] = {x, y};
    bool hit[NI];
    uint32_t local[NI];
    bool any = false;
#pragma unroll
    for (int p = 0; p < (NI + 1) / 2; ++p) {
        const uint32_t big = quad_big_nibbles(w[p]);
        const uint32_t b11 = HOT ? quad_base11_halves(w[p]) : 0u;
#pragma unroll
        for (int h = 0; h < 2 && 2 * p + h < NI; ++h) {
            const int j = 2 * p + h;
            const uint32_t big_h = h ? big >> 16 : big & 0xFFFFu, idx = h ? w[p] >> 16 : w[p] & 0xFFFFu;
            local[j] = HOT ? (h ? b11 >> 16 : b11 & 0xFFFFu) : idx - lo16;
            hit[j] = valid && (HOT ? big_h == 0u : (big_h != 0u && local[j] < QUAD_HOT));
            any |= hit[j];
            nhit_lane += hit[j] ? 1u : 0u;
        }
    }
    if (any) {
        const unsigned long long fixed = packed_add(dw, scale, cbits);
#pragma unroll
        for (int j = 0; j < NI; ++j)
            if (hit[j]) atomicAdd(reinterpret_cast<unsigned long long*>(acc) + local[j], fixed);
    }
    if (FB && valid) {
#pragma unroll
        for (int j = 0; j < NI; ++j)
            if (!hit[j]) {
                const uint32_t idx = (j & 1) ? w[j >> 1] >> 16 : w[j >> 1] & 0xFFFFu;
                const uint32_t rel = HOT ? QUAD_HOT + idx : quad_place(idx), ch = rel >> sl.cshift;       // (a miss of chunk 0 has a cell >= 11)
                if ((sl.fb_mask >> ch) & 1u) {
                    fb_add(D, Dc, sl.orb_dlo + rel, dw);
                    atomicAdd(&fb_hits[ch], 1u);
                }
            }
    }
}

// KIND (four-cell orbits, V < 5): 1 = chunk 0 of the hot-first order, 2 = one of the chunks behind it, 3 = plain order (QuadOrder); else 0
template <int N, int V, bool FB, bool FIXED, int KIND, bool STAG>
__device__ __forceinline__ void own_run(float* acc, const Slice& s, const TdRecs& recs, uint32_t B, uint32_t* hits, float* D, float* Dc,
                                        uint32_t* fb_hits, float scale, uint32_t cbits) {
    constexpr int F0 = OwnVariants<N>::f0(V), FC = OwnVariants<N>::fc(V);
    constexpr uint32_t IMAGES = N >= 4 ? COSET_MASK[V < 6 ? V : 0] : 0xFFu;
    uint32_t nhit = 0, nhit_wave = 0;
    if constexpr (N >= 4) {
        // main records from their precomputed orbit indices: 8 B (cross: 16 B, centre square: 2 B) + dw per record
        constexpr int NI = V == 4 ? 1 : 4, U = G2048_OWN_U;
        const OrbitIdx oi = orbit_idx(recs.oidx, B);
        const uint32_t lo_rel = s.tlo - s.orb_tlo;
        // a part takes every nparts-th block of records, not one contiguous range: the lanes may be ordered by board
        // pattern (LaneSort), and neighbouring records then hit the same few slots
        // (XCD-resident plan: the blocks b = x (mod 8) belong to XCD x for EVERY chunk, so that one L2 serves all their scans)
        constexpr uint32_t BLK = OWN_WG * U;
        const uint32_t nblk = (B + BLK - 1) / BLK, end = B;
        const uint32_t per_xcd = s.xcd ? s.nparts >> 3 : 0u;
        const uint32_t first = s.xcd ? s.part / per_xcd + 8u * (s.part % per_xcd) : s.part, stride = s.xcd ? 8u * per_xcd : s.nparts;
        // (two blocks per turn: the second one's 4 loads are out before the first one's adds — owner 0.0727 -> 0.0708 ms; three: no better.
        // The loop has to stay this plain for it: a hand-out of blocks through a bit pattern cost 10 %, profiles/r03_experiments.txt item 19)
        // Every workgroup starts its round at a block of its own (a hash of its number) and wraps: the scanners of all chunks walk
        // the same record blocks of their XCD, and walking them in step they all ask the same few L2 channels for the same lines at
        // the same time.  Staggered: owner 0.062 -> 0.058 ms, step -1.3 ... -2.3 % in four A/Bs (profiles/r04_experiments.txt item 14).
        auto block = [&](uint32_t base0) {
            uint32_t idx[U][NI];            // (V < 5: the record's index words as stored, idx[u][0 .. 1])
            float dw[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t r = base0 + threadIdx.x + (uint32_t)u * OWN_WG;
                const bool ok = r < end;
                const uint32_t rr = ok ? r : end - 1;
                if constexpr (V < 4) {
                    const uint2 t = oi.q[(size_t)V * B + rr];
                    idx[u][0] = t.x; idx[u][1] = t.y;
                } else if constexpr (V == 4) {
                    idx[u][0] = oi.c[rr];
                } else {
                    const uint4 t = oi.x[rr];
                    idx[u][0] = t.x; idx[u][1] = t.y; idx[u][2] = t.z; idx[u][3] = t.w;
                }
                const float d = recs.dw1[rr];
                dw[u] = ok ? d : 0.0f;
                if (recs.unit) dw[u] = dw[u] != 0.0f ? 1.0f : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if constexpr (V < 5 && KIND == 3) {         // plain order: the indices as they are, QUAD_HOT slots into the table
                    uint32_t plain[NI];
                    plain[0] = NI > 1 ? idx[u][0] & 0xFFFFu : idx[u][0];
                    if constexpr (NI > 1) { plain[1] = idx[u][0] >> 16; plain[2] = idx[u][1] & 0xFFFFu; plain[3] = idx[u][1] >> 16; }
                    own_accum_idx<NI, FB, FIXED, QUAD_HOT>(plain, dw[u], dw[u] != 0.0f, acc, s, lo_rel - QUAD_HOT, nhit_wave, D, Dc, fb_hits, (double)scale, cbits);
                } else if constexpr (V < 5) {
                    static_assert(V >= 5 || FIXED, "the four-cell orbits sum in fixed point (chunks of QUAD_HOT slots)");
                    own_accum_quad<NI, FB, KIND == 1>(idx[u][0], idx[u][NI > 1 ? 1 : 0], dw[u], dw[u] != 0.0f, acc, s, lo_rel - QUAD_HOT, nhit_wave, D, Dc,
                                                      fb_hits, (double)scale, cbits);
                } else {
                    own_accum_idx<NI, FB, FIXED>(idx[u], dw[u], dw[u] != 0.0f, acc, s, lo_rel, nhit_wave, D, Dc, fb_hits, (double)scale, cbits);
                }
            }
        
        };
        if constexpr (STAG) {       // (the hot-first kernel; the plain-order kernel keeps round 3's loop — a trained agent's step gains nothing from the stagger)
            const uint32_t nmine = first < nblk ? (nblk - first + stride - 1) / stride : 0u;
            const uint32_t start = first + (nmine ? ((blockIdx.x * 2654435761u) >> 16) % nmine : 0u) * stride;
            // (two loops of round 3's form, not one counted loop with a wrapping index: that one doubled the kernel's code and
            // spilled 1 800 scalar registers into VGPR lanes)
            _Pragma("unroll 2") for (uint32_t blk = start; blk < nblk; blk += stride) block(blk * BLK);
            _Pragma("unroll 2") for (uint32_t blk = first; blk < start; blk += stride) block(blk * BLK);
        } else {
            _Pragma("unroll 2") for (uint32_t blk = first; blk < nblk; blk += stride) block(blk * BLK);
        }
    } else {   // main records: this part's share of the lanes, OWN_UNROLL records per thread in flight; the loop bounds are
        // wave-uniform
        constexpr int OWN_UNROLL = 2;
        constexpr uint32_t BLK = OWN_WG * OWN_UNROLL;
        const uint32_t nblk = (B + BLK - 1) / BLK, end = B;
        for (uint32_t blk = s.part; blk < nblk; blk += s.nparts) {
            const uint32_t base0 = blk * BLK;
            uint4 st[OWN_UNROLL];
            float dw[OWN_UNROLL];
#pragma unroll
            for (int u = 0; u < OWN_UNROLL; ++u) {
                const uint32_t r = base0 + threadIdx.x + (uint32_t)u * OWN_WG;
                const bool ok = r < end;
                const uint32_t rr = ok ? r : end - 1;
                st[u] = recs.state1[rr];
                dw[u] = ok ? recs.dw1[rr] : 0.0f;
                if (recs.unit) dw[u] = dw[u] != 0.0f ? 1.0f : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < OWN_UNROLL; ++u) own_accum_small<N, V>(unpack4(st[u]), dw[u], dw[u] != 0.0f, acc, nhit, scale, cbits);
        }
    }
    {   // terminal queue
        const uint32_t q = *recs.qcount;
        const uint32_t begin = (uint32_t)((uint64_t)q * s.part / s.nparts), end = (uint32_t)((uint64_t)q * (s.part + 1) / s.nparts);
        for (uint32_t base0 = begin; base0 < end; base0 += OWN_WG) {
            const uint32_t r = base0 + threadIdx.x;
            const bool ok = r < end;
            const uint32_t rr = ok ? r : end - 1;
            if constexpr (N >= 4)
                own_accum<N, F0, FC, FB, FIXED, IMAGES, KIND == 3>(ld_packed(recs.qstate, rr), recs.unit ? 1.0f : recs.qdw[rr], ok, acc, s, nhit, D, Dc, fb_hits, scale,
                                                        cbits);
            else
                own_accum_small<N, V>(ld_packed(recs.qstate, rr), recs.unit ? 1.0f : recs.qdw[rr], ok, acc, nhit, scale, cbits);
        }
    }
    // load statistics for the planner: one counter bump per wave
    nhit += nhit_wave;              // (both are per-lane counts)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) nhit += __shfl_down(nhit, off);
    if ((threadIdx.x & 63) == 0 && nhit) atomicAdd(&hits[s.chunk], nhit);
}

// PLAIN: the kernel instance for the four-cell orbits' plain order (QuadOrder).  The two orders are two kernels, not one with a
// run-time switch: with every form of the record loop in one function the scalar registers no longer fit (7 500 v_readlane in the
// code object, 575 inside the loop that carries the fallback duty — a trained agent's busiest workgroups), with two they do.
template <int N, int V, bool PLAIN>
__device__ __forceinline__ void own_dispatch(float* acc, const Slice& s, const TdRecs& recs, uint32_t B, uint32_t* hits, float* D, float* Dc,
                                             uint32_t* fb_hits, float scale, uint32_t cbits) {
    if constexpr (V < OwnVariants<N>::COUNT) {
        if (s.variant == (uint32_t)V) {
            // the fallback duty (global atomics for chunks nobody holds in LDS) is carried by few workgroups: two
            // instantiations keep its tests out of everybody else's inner loop.  A variant that own_fixed leaves in fp32
            // (the cross orbit) also exists in fixed point: the one-pass mean rule needs the packed counts.
            const bool fb = N >= 4 && s.fb_mask;
            if constexpr (N >= 4 && V < 5) {       // a four-cell orbit: chunk 0 (all cells <= 10) or one of the four behind it
                static_assert(own_fixed(N, V) && FIXED_SLOTS == QUAD_HOT, "four-cell orbits: fixed point, chunks of QUAD_HOT slots");
                if constexpr (PLAIN) {
                    if (fb)
                        own_run<N, V, true, true, 3, !PLAIN>(acc, s, recs, B, hits, D, Dc, fb_hits, scale, cbits);
                    else
                        own_run<N, V, false, true, 3, !PLAIN>(acc, s, recs, B, hits, D, Dc, fb_hits, scale, cbits);
                } else if (s.tlo == s.orb_tlo) {
                    if (fb)
                        own_run<N, V, true, true, 1, !PLAIN>(acc, s, recs, B, hits, D, Dc, fb_hits, scale, cbits);
                    else
                        own_run<N, V, false, true, 1, !PLAIN>(acc, s, recs, B, hits, D, Dc, fb_hits, scale, cbits);
                } else {
                    if (fb)
                        own_run<N, V, true, true, 2, !PLAIN>(acc, s, recs, B, hits, D, Dc, fb_hits, scale, cbits);
                    else
                        own_run<N, V, false, true, 2, !PLAIN>(acc, s, recs, B, hits, D, Dc, fb_hits, scale, cbits);
                }
            } else if (own_fixed(N, V) || s.fixed) {
                if (fb)
                    own_run<N, V, true, true, 0, !PLAIN>(acc, s, recs, B, hits, D, Dc, fb_hits, scale, cbits);
                else
                    own_run<N, V, false, true, 0, !PLAIN>(acc, s, recs, B, hits, D, Dc, fb_hits, scale, cbits);
            } else if constexpr (!own_fixed(N, V)) {
                if (fb)
                    own_run<N, V, true, false, 0, !PLAIN>(acc, s, recs, B, hits, D, Dc, fb_hits, scale, cbits);
                else
                    own_run<N, V, false, false, 0, !PLAIN>(acc, s, recs, B, hits, D, Dc, fb_hits, scale, cbits);
            }
        } else
            own_dispatch<N, V + 1, PLAIN>(acc, s, recs, B, hits, D, Dc, fb_hits, scale, cbits);
    }
}

// `dst` is the orbit table D (n >= 4) or the weight table itself (n = 2, 3).  `cdst` (null unless the per-slot mean rule runs
// in one pass): where the add counts go — the accumulation then packs count and sum into one 64-bit LDS word.
template <int N, bool PLAIN>
__global__ __launch_bounds__(OWN_WG) void k_td_update_owner(float* dst, float* cdst, TdRecs recs, uint32_t B, const Slice* slices, uint32_t* hits,
                                                            uint64_t* wg_clock) {
    __shared__ __attribute__((aligned(16))) float acc[OWN_SLOTS];
    __shared__ uint32_t fb_hits[64];
    const Slice s = slices[blockIdx.x];
    if (threadIdx.x == 0) wg_clock[2 * blockIdx.x] = wall_clock64();    // feeds the planner; g2048_debug_owner_plan shows them
    const bool fixed = own_fixed(N, (int)s.variant) || s.fixed;
    // fixed-point scale 2^S from the step's largest |dw| (< 2^e): a slot can take every visited image of every record
    // of this part (2^add_bits adds), and |sum| < 2^(add_bits + e + S) must fit the sum field (61 bits, or what the count
    // bits leave of them)
    float scale = 1.0f, inv_scale = 1.0f;
    uint32_t cbits = 0;
    if (fixed) {
        const float big = recs.unit ? 1.0f : __uint_as_float(*recs.dwmax);
        int e = big > 0.0f ? ilogbf(big) + 1 : 0;
        const uint32_t part_recs = (B + s.nparts - 1) / s.nparts + 4096u + *recs.qcount;      // (blocks of <= 4096 records, dealt round-robin)
        const uint32_t per_rec = N >= 4 ? 4u : 8u;          // adds one record can make to one slot
        const int add_bits = 32 - __clz((int)(part_recs < (1u << 28) ? per_rec * part_recs : 0x7FFFFFFFu));
        if (cdst) cbits = (uint32_t)add_bits + 1u;
        int S = 38 - e;
        if (S > 61 - (int)cbits - e - add_bits) S = 61 - (int)cbits - e - add_bits;
        S = S > 100 ? 100 : (S < -60 ? -60 : S);
        scale = ldexpf(1.0f, S);
        inv_scale = ldexpf(1.0f, -S);
    }
    const uint32_t words = fixed ? 2 * s.size : s.size;        // a fixed-point slot is two LDS words
    {   // (16 bytes per store: 2.0 -> 1.x us per workgroup; words is a multiple of 4)
        float4* const acc4 = reinterpret_cast<float4*>(acc);
        for (uint32_t j = threadIdx.x; j < words / 4u; j += OWN_WG) acc4[j] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    if (threadIdx.x < 64) fb_hits[threadIdx.x] = 0;
    __syncthreads();
    own_dispatch<N, 0, PLAIN>(acc, s, recs, B, hits, dst, cdst, fb_hits, scale, cbits);
    __syncthreads();
    if (threadIdx.x < 64 && fb_hits[threadIdx.x]) atomicAdd(&hits[s.chunk0 + threadIdx.x], fb_hits[threadIdx.x]);
    const unsigned long long cmask = (1ull << cbits) - 1ull;
    for (uint32_t j = threadIdx.x; j < s.size; j += OWN_WG) {
        float v, cnt = 0.0f;
        if (fixed) {
            const unsigned long long word = reinterpret_cast<const unsigned long long*>(acc)[j];
            if (word == 0ull) continue;                                        // (untouched: most slots of most chunks)
            const unsigned long long n = word & cmask;                         // (0 without count bits)
            v = (float)((double)((long long)(word - n) >> cbits) * (double)inv_scale);
            cnt = (float)n;
        } else {
            v = acc[j];
        }
        if (s.nparts == 1) {                                   // this workgroup is the only writer of the slice
            if (v != 0.0f) dst[s.dlo + j] += v;
            if (cnt != 0.0f) cdst[s.dlo + j] += cnt;
        } else {
            if (v != 0.0f) __hip_atomic_fetch_add(&dst[s.dlo + j], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cnt != 0.0f) __hip_atomic_fetch_add(&cdst[s.dlo + j], cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (threadIdx.x == 0) wg_clock[2 * blockIdx.x + 1] = wall_clock64();
}

// The planner's statistics (hit counters + workgroup clocks of the owner kernel that just ran, 17 KB) go to their pinned
// host mirror from inside the apply kernel — the first blocks store them over PCIe — instead of through a copy command:
// a blit kernel on a side stream took 11-17 us beside every k_td_play (rocprofv3, round 2).
// The host learns that they have arrived from a sequence number the same block stores behind them (late round 3): an event
// recorded between this kernel and the next step's k_td_play is a marker packet in the queue, and the command processor spent
// 6 us on it every step before it dispatched k_td_play (rocprofv3 trace: the only gap in the step's chain).
struct StatMirror {
    const uint32_t* src;    // device: statbuf
    uint32_t* dst;          // pinned host memory (device-visible); null: no mirror this step
    uint32_t words;
    uint32_t seq;           // stored at dst[words] once the statistics are in host memory; dst[words + 1] = their checksum
    unsigned long long* done;   // device: {checksum so far : 32 | mirroring blocks finished : 32} of THIS launch (the last block clears it)
};
// The first mirror_blocks() blocks of an apply launch do nothing else (the system-scope fence waits for the stores' trip over
// PCIe, ~2 us: inside a block that also has table work it lengthened the kernel by that much); the others see their index shifted.
__host__ __device__ __forceinline__ uint32_t mirror_blocks(const StatMirror& m) { return m.dst ? (m.words + WG - 1) / WG : 0u; }
__device__ __forceinline__ void mirror_stats(const StatMirror& m) {
    const uint32_t nblk = mirror_blocks(m);     // one word per thread
    if (blockIdx.x >= nblk) return;
    __shared__ uint32_t part;
    if (threadIdx.x == 0) part = 0;
    __syncthreads();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t v = 0;
    if (i < m.words) {
        v = __hip_atomic_load(&m.src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // (written by the previous kernel: bypass a stale L1 line)
        m.dst[i] = v;
    }
    uint32_t sum = v;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off);
    if ((threadIdx.x & 63u) == 0) atomicAdd(&part, sum);
    // The stores go straight to pinned, uncached host memory.  What has to stand between them and the announcement is that they
    // have been PERFORMED: every storing wave waits for its stores' acknowledgements here (a workgroup-scope fence emits no wait
    // on gfx950 — round 3 had only that, the advisor's finding; a system-scope release fence would also write back and
    // invalidate the XCD's L2 in the middle of the step, +1.2 us, profiles/r03_experiments.txt item 23b).  The wait orders the
    // hardware; for whatever the memory model leaves open the set carries a checksum, which the host verifies before it
    // trusts the statistics (mirror_arrived).  tools/check_codeobj.py checks that this wait is in the code object.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // The block that finishes last announces the set.  One 64-bit atomic carries the block's checksum part and its "done" tick,
    // so the last block sees every part; it is issued behind the wait above (in-order issue, "memory" clobber for the compiler).
    if (threadIdx.x == 0) {
        const unsigned long long old = atomicAdd(m.done, ((unsigned long long)part << 32) | 1ull);
        if ((uint32_t)old == nblk - 1u) {
            const uint32_t total = (uint32_t)(old >> 32) + part;
            *m.done = 0ull;          // (nobody else touches it before the next mirroring launch, which is behind this one on the stream)
            // checksum first, sequence word second: both uncached stores of one thread to one 8-byte-aligned pair, issued as ONE store
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(m.dst + m.words), (unsigned long long)total << 32 | m.seq, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// table_i[perm_i(k)] += v for every member i of the orbit; `dacc` (may be null) mirrors the add (g2048_delta_begin)
__device__ __forceinline__ void add_to_members(float* w, float* dacc, const OrbitInfo& oi, uint32_t k, float v) {
    // The members are different features: their slots are distinct, so all the loads go out before the first store (written as
    // `w[slot] += v` in a loop the compiler has to keep in order, eight dependent round trips per thread; worth 0.5 - 0.8 us per launch).
    uint32_t slot[MAX_MEMBERS];
    float old[MAX_MEMBERS], dold[MAX_MEMBERS];
#pragma unroll
    for (uint32_t m = 0; m < MAX_MEMBERS; ++m) {
        if (m >= oi.nmem) break;
        // (the four- and five-cell tables of n >= 4 live in table_place order, the f_6 tables in hex_place order, n = 2, 3 in index order)
        slot[m] = oi.radix != 16u ? oi.off[m] + permute_hex_placed(k, oi.perm[m])
                                  : oi.digits >= 4u ? table_place(oi.off[m] + permute_digits(k, oi.perm[m], oi.digits, 16u))
                                                    : oi.off[m] + permute_digits(k, oi.perm[m], oi.digits, 16u);
        old[m] = w[slot[m]];
        if (dacc) dold[m] = dacc[slot[m]];
    }
#pragma unroll
    for (uint32_t m = 0; m < MAX_MEMBERS; ++m) {
        if (m >= oi.nmem) break;
        w[slot[m]] = old[m] + v;
        if (dacc) dacc[slot[m]] = dold[m] + v;
    }
}

// D -> every member table of its orbit (plain read-modify-write: for one member the permutation is a bijection, and
// members are different features, so no two threads touch the same slot), then D is cleared for the next step.
// The LDS-owned orbit tables [0, owned) are double-buffered (cur: this step's sums, oth: the next step's, cleared here;
// a clear in place would race with the threads that read E[sigma(k)]; a memset between the steps costs a launch and, in
// ROCclr, ~20 us of idle queue); the f_6 orbit tables behind them live in D and are cleared in place.
__global__ __launch_bounds__(WG) void k_apply_orbits(float* w, float* dacc, float* D, const float* cur, float* oth, uint32_t owned, OrbitTable t,
                                                      StatMirror sm, uint32_t quad_plain) {
    mirror_stats(sm);
    if (blockIdx.x < mirror_blocks(sm)) return;
    const uint32_t K = (blockIdx.x - mirror_blocks(sm)) * WG + threadIdx.x;
    if (K >= t.total) return;
    if (K < owned) oth[K] = 0.0f;
    uint32_t o = 0;
#pragma unroll
    for (uint32_t j = 1; j < MAX_ORBITS; ++j)
        if (j < t.count && K >= t.o[j].base) o = j;
    const OrbitInfo& oi = t.o[o];
    const bool cross = oi.radix == 16u && oi.digits == 5u, quad = oi.radix == 16u && oi.digits == 4u;    // tables in cross_order / quad_place order
    uint32_t k = cross ? cross_unorder(K - oi.base) : K - oi.base;
    if (quad && !quad_unplace(K - oi.base, k, quad_plain != 0u)) return;                  // (a hole of the four-cell order)
    float v;
    uint32_t k2 = k;        // f_6 orbit with a stabiliser {e, sigma}: the thread of the smaller of k, sigma(k) serves both
    if (K >= owned) {
        v = D[K];
        if (oi.nstab == 2) {
            k2 = permute_digits(k, oi.sperm[1], oi.digits, oi.radix);
            if (k2 < k) return;
            v += D[oi.base + k2];
            if (v == 0.0f) return;
            D[oi.base + k2] = 0.0f;
        } else if (v == 0.0f) {
            return;
        }
        D[K] = 0.0f;
    } else {            // symmetrise over the stabiliser
        v = cur[K];
        for (uint32_t s = 1; s < oi.nstab; ++s) {
            const uint32_t j = permute_digits(k, oi.sperm[s], oi.digits, oi.radix);
            v += cur[oi.base + (cross ? cross_order(j) : quad ? (quad_plain ? QUAD_HOT + j : quad_place(j)) : j)];
        }
        if (v == 0.0f) return;
    }
    add_to_members(w, dacc, oi, k, v);
    if (k2 != k) add_to_members(w, dacc, oi, k2, v);
}

// Per-slot mean rule (g2048_set_update_rule): S = sum of the dw that target a slot, C = how many did; the slot moves
// by S / C.  S and C come from two runs of the same accumulation (the second with dw = 1).
__global__ __launch_bounds__(WG) void k_apply_orbits_mean(float* w, float* dacc, float* S, float* C, const float* scur, const float* ccur, float* soth,
                                                          float* coth, uint32_t owned, OrbitTable t, StatMirror sm, uint32_t quad_plain) {
    mirror_stats(sm);
    if (blockIdx.x < mirror_blocks(sm)) return;
    const uint32_t K = (blockIdx.x - mirror_blocks(sm)) * WG + threadIdx.x;
    if (K >= t.total) return;
    if (K < owned) {
        soth[K] = 0.0f;
        coth[K] = 0.0f;
    }
    uint32_t o = 0;
#pragma unroll
    for (uint32_t j = 1; j < MAX_ORBITS; ++j)
        if (j < t.count && K >= t.o[j].base) o = j;
    const OrbitInfo& oi = t.o[o];
    const bool cross = oi.radix == 16u && oi.digits == 5u, quad = oi.radix == 16u && oi.digits == 4u;
    uint32_t k = cross ? cross_unorder(K - oi.base) : K - oi.base;
    if (quad && !quad_unplace(K - oi.base, k, quad_plain != 0u)) return;
    float cnt, sum;
    uint32_t k2 = k;
    if (K >= owned) {
        cnt = C[K];
        sum = S[K];
        if (oi.nstab == 2) {
            k2 = permute_digits(k, oi.sperm[1], oi.digits, oi.radix);
            if (k2 < k) return;
            cnt += C[oi.base + k2];
            sum += S[oi.base + k2];
            if (cnt == 0.0f) return;
            S[oi.base + k2] = 0.0f;
            C[oi.base + k2] = 0.0f;
        } else if (cnt == 0.0f) {
            return;
        }
        S[K] = 0.0f;
        C[K] = 0.0f;
    } else {
        cnt = ccur[K];
        sum = scur[K];
        for (uint32_t s = 1; s < oi.nstab; ++s) {
            const uint32_t pj = permute_digits(k, oi.sperm[s], oi.digits, oi.radix);
            const uint32_t j = oi.base + (cross ? cross_order(pj) : quad ? (quad_plain ? QUAD_HOT + pj : quad_place(pj)) : pj);
            cnt += ccur[j];
            sum += scur[j];
        }
        if (cnt == 0.0f) return;
    }
    const float v = sum / cnt;
    add_to_members(w, dacc, oi, k, v);
    if (k2 != k) add_to_members(w, dacc, oi, k2, v);
}

// n = 6: the twelve 14^6-slot tables (361 MB) do not fit in LDS, so their adds end as global atomics — but (1) through
// the orbits: the 12 features fall into two orbits (8 corner blocks, 4 middle blocks) whose representatives are
// features 21 and 22, so a record costs 2 x 8 = 16 adds into D instead of 96 into the table; and (2) through a
// per-workgroup combining cache in LDS: memory-side atomics serialise per address (a slot that takes 5 % of a step's
// adds costs milliseconds), so every workgroup first sums its adds in a 16 384-entry direct-mapped {slot, sum} cache
// and only cache misses and the final flush touch HBM.
constexpr uint32_t TAIL_CACHE = 16384, TAIL_EMPTY = 0xFFFFFFFFu;

__device__ __forceinline__ void cached_add(uint32_t* keys, float* vals, float* D, uint32_t slot, float dw) {
    const uint32_t h = (slot * 2654435761u) >> 18;                 // 14 bits
    const uint32_t prev = atomicCAS(&keys[h], TAIL_EMPTY, slot);
    if (prev == TAIL_EMPTY || prev == slot)
        atomicAdd(&vals[h], dw);
    else
        __hip_atomic_fetch_add(&D[slot], dw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int N>
__global__ __launch_bounds__(OWN_WG) void k_td_update_tail(float* D, TdRecs recs, uint32_t B, uint32_t dbaseA, uint32_t dbaseB) {
    constexpr int F = Shape<N>::F;
    __shared__ uint32_t keys[TAIL_CACHE];
    __shared__ float vals[TAIL_CACHE];
    for (uint32_t j = threadIdx.x; j < TAIL_CACHE; j += OWN_WG) {
        keys[j] = TAIL_EMPTY;
        vals[j] = 0.0f;
    }
    __syncthreads();
    const uint32_t total = (B + *recs.qcount) * 8u;                // (record, image) pairs; records >= B are the terminal queue
    for (uint32_t t = blockIdx.x * OWN_WG + threadIdx.x; t < total; t += gridDim.x * OWN_WG) {
        const uint32_t r = t >> 3, g = t & 7u;
        Packed p;
        float dw;
        if (r < B) {
            dw = recs.dw1[r];
            if (dw == 0.0f) continue;
            p = ld_packed(recs.state1, r);
        } else {
            dw = recs.qdw[r - B];
            p = ld_packed(recs.qstate, r - B);
        }
        if (recs.unit) dw = 1.0f;
        uint32_t s[F];
        feature_slots<N>(d4_image(p, g), s);
        cached_add(keys, vals, D, dbaseA + (s[21] - feature_offset(N, 21)), dw);
        if ((HEX_COSET_MASK[1] >> g) & 1u) cached_add(keys, vals, D, dbaseB + (s[22] - feature_offset(N, 22)), dw);     // one image per coset
    }
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < TAIL_CACHE; j += OWN_WG)
        if (keys[j] != TAIL_EMPTY && vals[j] != 0.0f) __hip_atomic_fetch_add(&D[keys[j]], vals[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------------------------------------------------ n = 6: binned update of the f_6 orbits
// The two f_6 orbit tables (2 x 14^6 = 15 059 072 slots of D behind the LDS-owned ones) take 8 + 4 adds per record.
// k_td_update_tail sends them to memory as scattered atomics (~16 G/s: 0.75 ms per step of 2^20 lanes).  Here they are
// first BINNED by table chunk and then summed by a workgroup that owns the chunk in LDS, like the other orbits:
//   k_hex_scatter  every tile of 1 024 records: the records' 16 (slot, dw) pairs, counted per chunk in LDS; room for the tile's
//                  share of a chunk's run reserved with ONE atomic per chunk it touches; the pairs put in chunk order in LDS and
//                  copied out coalesced
//   k_hex_plan     how many pairs every chunk got -> a work list of (chunk, part) for the owners; and the NEXT step's layout
//   k_hex_owner    one workgroup per (chunk, part): pairs -> 64-bit fixed-point LDS sums (counts packed under them for the
//                  mean rule) -> D (one writer per slot unless the chunk was split)
// Round 4: there is no counting pass any more (round 3 ran the slot computation twice: k_hex_count, 26 us, VALU-bound).  The
// runs of step t are laid out from what the chunks received in step t - 1 — cap = demand x 1.25 + 1 024 pairs, the board
// distribution moves slowly — and a pair that finds its chunk's run full is added to D with a global atomic on the spot
// (young, synchronised boards, whose distribution does move fast, pay that for their first steps; the sums are the same).
// k_apply_orbits then moves D into the twelve member tables as before.
constexpr uint32_t HEX_SLOTS = 2u * HEX_SIZE;               // both orbit tables, contiguous in D
constexpr uint32_t HEX_CHUNK = 16384u;                      // slots per chunk: 128 KB of 64-bit LDS sums
constexpr uint32_t HEX_BINS = (HEX_SLOTS + HEX_CHUNK - 1) / HEX_CHUNK;     // 920
constexpr uint32_t HEX_PART_PAIRS = 1u << 16;               // an owner workgroup takes at most this many pairs
constexpr uint32_t HEX_MAX_WORK = 4096;
constexpr uint32_t HEX_YOUNG_STEPS = 64;        // TD steps after a (re)start that count their pairs before they place them (k_hex_count)

struct HexWork {        // one owner workgroup
    uint32_t bin, first, count, split;      // pairs [first, first + count) of chunk `bin`; split: the chunk has several workgroups
};

struct HexBufs {
    uint32_t* count;        // [HEX_BINS] capacity of each chunk's run (this step)
    uint32_t* base;         // [HEX_BINS + 1] first pair of each chunk's run
    uint32_t* cursor;       // [HEX_BINS] next free pair of each run during the scatter (may pass the run's end: what did went to D as atomics)
    uint2* pairs;           // (slot in the two tables, dw bits)
    HexWork* work;          // [HEX_MAX_WORK]
    uint32_t* nwork;
    uint32_t* touched;      // [HEX_BINS] pairs the chunk was sent this step (0: k_hex_apply skips it)
    uint32_t pairs_cap;     // entries of `pairs`
};

// the 16 slots (relative to the first f_6 orbit table) one record adds to: all 8 images of the corner-block representative
// and of the middle-block representative.  (The middle block is fixed by the left-right mirror, and k_td_update_tail visits
// one image per coset and lets k_apply_orbits pair k with sigma(k); here the pairs are cheap and the apply step stays local
// to a chunk, so all 8 images are visited and D is the orbit table itself.)
constexpr int HEX_PAIRS = 16;
template <int N>
__device__ __forceinline__ void hex_record_slots(const Packed& p, uint32_t (&out)[HEX_PAIRS]) {
    // The two orbit tables are indexed in hex_place order, like the member tables in memory: boards with small tiles then
    // fall into a few dozen of the 920 chunks instead of a few hundred, which is what the count / scatter / apply passes pay for.
#pragma unroll
    for (uint32_t g = 0; g < 8; ++g) {
        uint32_t s[12];
        hex_slots_placed(d4_image(p, g), s);            // (only the two representatives' entries survive dead-code elimination)
        out[g] = s[0] - HEX_BASE;
        out[8 + g] = HEX_SIZE + (s[1] - HEX_BASE - HEX_SIZE);
    }
}

// record r of the step: main records 0 .. B-1 (dw1 != 0), then the terminal queue
__device__ __forceinline__ bool hex_load_record(const TdRecs& recs, uint32_t B, uint32_t r, uint32_t total, Packed& p, float& dw) {
    if (r >= total) return false;
    if (r < B) {
        dw = recs.dw1[r];
        if (dw == 0.0f) return false;
        p = ld_packed(recs.state1, r);
    } else {
        dw = recs.qdw[r - B];
        p = ld_packed(recs.qstate, r - B);
    }
    return true;
}

// the first layout of a context: equal runs (nothing is known yet; what does not fit goes to D as atomics)
__global__ __launch_bounds__(OWN_WG) void k_hex_layout_init(HexBufs hb) {
    const uint32_t j = threadIdx.x, per = hb.pairs_cap / HEX_BINS;
    if (j < HEX_BINS) {
        hb.count[j] = per;
        hb.base[j] = j * per;
        hb.cursor[j] = j * per;
        hb.touched[j] = 0;
    }
    if (j == 0) hb.base[HEX_BINS] = HEX_BINS * per;
}

// YOUNG BOARDS.  For the first steps after g2048_create / g2048_reset every lane is in its opening: a handful of chunks takes all
// 16 x 2^20 pairs and the handful changes from step to step, so "last step's demand" predicts nothing and the pairs that find their
// run full would go to D as same-address global atomics (measured: 88 ms for the second step of a fresh 2^20-lane context, 4.5, 2.1,
// 0.6 ms for the next ones, back to 60 us after ~48 steps).  Those steps therefore get the exact layout round 3 always computed:
// count every chunk's pairs first (this kernel), then lay the runs out to fit (k_hex_layout_exact).
template <int N>
__global__ __launch_bounds__(OWN_WG) void k_hex_count(TdRecs recs, uint32_t B, uint32_t* demand) {
    __shared__ uint32_t hist[HEX_BINS];
    for (uint32_t j = threadIdx.x; j < HEX_BINS; j += OWN_WG) hist[j] = 0;
    __syncthreads();
    const uint32_t total = B + *recs.qcount;
    for (uint32_t r = blockIdx.x * OWN_WG + threadIdx.x; r < total; r += gridDim.x * OWN_WG) {
        Packed p;
        float dw;
        if (!hex_load_record(recs, B, r, total, p, dw)) continue;
        uint32_t s[HEX_PAIRS];
        hex_record_slots<N>(p, s);
#pragma unroll
        for (int j = 0; j < HEX_PAIRS; ++j) atomicAdd(&hist[s[j] / HEX_CHUNK], 1u);
    }
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < HEX_BINS; j += OWN_WG)
        if (hist[j]) atomicAdd(&demand[j], hist[j]);
}

__global__ __launch_bounds__(OWN_WG) void k_hex_layout_exact(HexBufs hb, const uint32_t* demand) {
    __shared__ uint32_t wtot[OWN_WG / 64];
    const uint32_t j = threadIdx.x, lane = j & 63u, wave = j >> 6;
    const uint32_t d = j < HEX_BINS ? demand[j] : 0u;
    uint32_t inc = d;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)inc, off);
        if (lane >= (uint32_t)off) inc += up;
    }
    if (lane == 63u) wtot[wave] = inc;
    __syncthreads();
    for (uint32_t k = 0; k < wave; ++k) inc += wtot[k];
    if (j < HEX_BINS) {
        hb.count[j] = d;
        hb.base[j] = inc - d;
        hb.cursor[j] = inc - d;
        if (j == HEX_BINS - 1) hb.base[HEX_BINS] = inc;
    }
}

// one workgroup, behind the scatter: what every chunk received -> the owners' work list (busy chunks cut into parts) -> the next
// step's layout.  Three inclusive scans side by side (parts, next capacities, plain demands): shuffles inside the waves, 16 totals
// across them.
__global__ __launch_bounds__(OWN_WG) void k_hex_plan(HexBufs hb) {
    __shared__ uint32_t wpre[OWN_WG], cnt[OWN_WG], first_of[OWN_WG], wtot[3][OWN_WG / 64];
    const uint32_t j = threadIdx.x, lane = j & 63u, wave = j >> 6;
    const uint32_t base = j < HEX_BINS ? hb.base[j] : 0u, cap = j < HEX_BINS ? hb.count[j] : 0u;
    const uint32_t demand = j < HEX_BINS ? hb.cursor[j] - base : 0u, stored = demand < cap ? demand : cap;
    const uint32_t parts = (stored + HEX_PART_PAIRS - 1) / HEX_PART_PAIRS;
    const uint32_t want = j < HEX_BINS ? demand + demand / 4u + 1024u : 0u;
    uint32_t ip = parts, iw = want, id = demand;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)ip, off), uw = (uint32_t)__shfl_up((int)iw, off), ud = (uint32_t)__shfl_up((int)id, off);
        if (lane >= (uint32_t)off) {
            ip += up;
            iw += uw;
            id += ud;
        }
    }
    if (lane == 63u) {
        wtot[0][wave] = ip;
        wtot[1][wave] = iw;
        wtot[2][wave] = id;
    }
    __syncthreads();
    uint32_t all_want = 0;
    for (uint32_t k = 0; k < OWN_WG / 64; ++k) {
        if (k < wave) {
            ip += wtot[0][k];
            iw += wtot[1][k];
            id += wtot[2][k];
        }
        all_want += wtot[1][k];
    }
    wpre[j] = ip;
    cnt[j] = stored;
    first_of[j] = base;
    __syncthreads();
    const uint32_t nwork = wpre[OWN_WG - 1] < HEX_MAX_WORK ? wpre[OWN_WG - 1] : HEX_MAX_WORK;
    if (j == 0) *hb.nwork = nwork;
    // work item w belongs to the chunk whose inclusive part count first exceeds w
    for (uint32_t w = j; w < nwork; w += OWN_WG) {
        uint32_t lo = 0, hi = OWN_WG - 1;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (wpre[mid] > w)
                hi = mid;
            else
                lo = mid + 1;
        }
        const uint32_t bin = lo, cb = cnt[bin], pb = (cb + HEX_PART_PAIRS - 1) / HEX_PART_PAIRS, k = w - (wpre[bin] - pb);
        const uint32_t plo = (uint32_t)((uint64_t)cb * k / pb), phi = (uint32_t)((uint64_t)cb * (k + 1) / pb);
        hb.work[w] = HexWork{bin, first_of[bin] + plo, phi - plo, pb > 1 ? 1u : 0u};
    }
    // the next step's runs: room for a quarter more than this step's demand (if that does not fit, for exactly the demand)
    const bool roomy = all_want <= hb.pairs_cap;
    if (j < HEX_BINS) {
        const uint32_t next_cap = roomy ? want : demand, next_base = roomy ? iw - want : id - demand;
        hb.touched[j] = demand;
        hb.count[j] = next_cap;
        hb.base[j] = next_base;
        hb.cursor[j] = next_base;
        if (j == HEX_BINS - 1) hb.base[HEX_BINS] = next_base + next_cap;
    }
}

// Round 4: the tile's pairs are first put in chunk order in LDS (128 KB: 1 024 records x 16 pairs) and then copied out, so that
// consecutive lanes write consecutive pairs of a chunk's run.  The round-3 form stored every pair straight from the lane that made
// it — sixteen 8-byte stores per lane into sixteen different runs, 15.3 M separate L1 accesses per launch with the L1 busy 90 % of
// the kernel's 64 us (TCP_TOTAL_CACHE_ACCESSES / TCP_GATE_EN1, profiles/r04_pmc_summary.json): a request-rate bound like k_td_play's.
template <int N>
__global__ __launch_bounds__(OWN_WG) void k_hex_scatter(TdRecs recs, uint32_t B, HexBufs hb, float* D, float* Dc) {
    __shared__ uint2 stage[OWN_WG * HEX_PAIRS];
    __shared__ uint32_t hist[HEX_BINS], lofs[HEX_BINS], lbase[HEX_BINS], lend[HEX_BINS], wtot[OWN_WG / 64];
    for (uint32_t j = threadIdx.x; j < HEX_BINS; j += OWN_WG) lend[j] = hb.base[j] + hb.count[j];      // where each chunk's run ends
    const uint32_t total = B + *recs.qcount;
    const uint32_t ntiles = (total + OWN_WG - 1) / OWN_WG;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {        // (uniform trip count: barriers inside)
        for (uint32_t j = threadIdx.x; j < HEX_BINS; j += OWN_WG) hist[j] = 0;
        __syncthreads();
        Packed p;
        float dw = 0.0f;
        uint32_t s[HEX_PAIRS], rank[HEX_PAIRS];
        const bool ok = hex_load_record(recs, B, tile * OWN_WG + threadIdx.x, total, p, dw);
        if (ok) {
            hex_record_slots<N>(p, s);
#pragma unroll
            for (int j = 0; j < HEX_PAIRS; ++j) rank[j] = atomicAdd(&hist[s[j] / HEX_CHUNK], 1u);      // place inside the tile's share of the chunk
        }
        __syncthreads();
        // one thread per chunk: room in the chunk's run, and the chunk's offset in the staging area (an
        // exclusive scan of the histogram: wave scans + the 16 wave totals)
        const uint32_t h = threadIdx.x < HEX_BINS ? hist[threadIdx.x] : 0u;
        uint32_t inc = h;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)inc, off);
            if (lane >= (uint32_t)off) inc += up;
        }
        if (lane == 63u) wtot[wave] = inc;
        if (threadIdx.x < HEX_BINS) lbase[threadIdx.x] = h ? atomicAdd(&hb.cursor[threadIdx.x], h) : 0u;
        __syncthreads();
        uint32_t before = 0;
        for (uint32_t k = 0; k < wave; ++k) before += wtot[k];
        if (threadIdx.x < HEX_BINS) lofs[threadIdx.x] = before + inc - h;
        uint32_t npairs = 0;
        for (uint32_t k = 0; k < OWN_WG / 64; ++k) npairs += wtot[k];
        __syncthreads();
        if (ok) {
            const uint32_t bits = recs.unit ? __float_as_uint(1.0f) : __float_as_uint(dw);
#pragma unroll
            for (int j = 0; j < HEX_PAIRS; ++j) stage[lofs[s[j] / HEX_CHUNK] + rank[j]] = make_uint2(s[j], bits);
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < npairs; i += OWN_WG) {
            const uint2 pr = stage[i];
            const uint32_t bin = pr.x / HEX_CHUNK, pos = lbase[bin] + (i - lofs[bin]);
            if (pos < lend[bin]) {
                hb.pairs[pos] = pr;
            } else {                        // the run is full (the chunk got more than 1.25 x last step's pairs): straight to D
                __hip_atomic_fetch_add(&D[pr.x], __uint_as_float(pr.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (Dc) __hip_atomic_fetch_add(&Dc[pr.x], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
    }
}

// D (and Dc: the add counts, when the mean rule runs in one pass) += the sums of one chunk part
__global__ __launch_bounds__(OWN_WG) void k_hex_owner(float* D, float* Dc, const TdRecs recs, HexBufs hb) {
    __shared__ unsigned long long acc[HEX_CHUNK];
    if (blockIdx.x >= *hb.nwork) return;
    const HexWork wk = hb.work[blockIdx.x];
    // fixed-point scale as in k_td_update_owner: |sum| < count x 2^(e + S) must fit beside the count bits
    const float big = recs.unit ? 1.0f : __uint_as_float(*recs.dwmax);
    const int e = big > 0.0f ? ilogbf(big) + 1 : 0;
    const int add_bits = 32 - __clz((int)(wk.count | 1u));
    const uint32_t cbits = Dc ? (uint32_t)add_bits + 1u : 0u;
    int S = 38 - e;
    if (S > 61 - (int)cbits - e - add_bits) S = 61 - (int)cbits - e - add_bits;
    S = S > 100 ? 100 : (S < -60 ? -60 : S);
    const double scale = (double)ldexpf(1.0f, S), inv_scale = (double)ldexpf(1.0f, -S);
    for (uint32_t j = threadIdx.x; j < HEX_CHUNK; j += OWN_WG) acc[j] = 0ull;
    __syncthreads();
    const uint32_t lo = wk.bin * HEX_CHUNK;
    // eight pairs per thread in flight: one load per turn left the 64 turns of a 65 536-pair part waiting for memory one after
    // the other (52 us for 16.8 M adds the LDS could take in 7)
    constexpr uint32_t HU = 8;
    for (uint32_t k0 = 0; k0 < wk.count; k0 += HU * OWN_WG) {
        uint2 pr[HU];
#pragma unroll
        for (uint32_t u = 0; u < HU; ++u) {
            const uint32_t k = k0 + u * OWN_WG + threadIdx.x;
            pr[u] = hb.pairs[wk.first + (k < wk.count ? k : wk.count - 1u)];
        }
#pragma unroll
        for (uint32_t u = 0; u < HU; ++u)
            if (k0 + u * OWN_WG + threadIdx.x < wk.count) atomicAdd(&acc[pr[u].x - lo], packed_add(__uint_as_float(pr[u].y), scale, cbits));
    }
    __syncthreads();
    const unsigned long long cmask = (1ull << cbits) - 1ull;
    for (uint32_t j = threadIdx.x; j < HEX_CHUNK; j += OWN_WG) {
        const unsigned long long word = acc[j];
        if (!word || lo + j >= HEX_SLOTS) continue;
        const unsigned long long n = word & cmask;
        const float v = (float)((double)((long long)(word - n) >> cbits) * inv_scale), cnt = (float)n;
        if (wk.split) {
            if (v != 0.0f) __hip_atomic_fetch_add(&D[lo + j], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cnt != 0.0f) __hip_atomic_fetch_add(&Dc[lo + j], cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (v != 0.0f) D[lo + j] += v;
            if (cnt != 0.0f) Dc[lo + j] += cnt;
        }
    }
}

// D -> the member tables of the two f_6 orbits, chunk by chunk, only for the chunks that received pairs this step (a
// step touches a few hundred of the 920); clears what it read.  Sc / Cc non-null: per-slot mean rule (v = S / C).
__global__ __launch_bounds__(WG) void k_hex_apply(float* w, float* dacc, float* Dh, float* Ch, HexBufs hb, OrbitInfo oa, OrbitInfo ob) {
    const uint32_t K = blockIdx.x * WG + threadIdx.x;           // (a workgroup's 256 slots lie in one chunk)
    const uint32_t bin = K / HEX_CHUNK;
    if (K >= HEX_SLOTS || hb.touched[bin] == 0) return;
    float v = Dh[K];
    if (Ch) {
        const float cnt = Ch[K];
        if (cnt == 0.0f) return;
        v /= cnt;
        Ch[K] = 0.0f;
    } else if (v == 0.0f) {
        return;
    }
    Dh[K] = 0.0f;
    if (K < HEX_SIZE)
        add_to_members(w, dacc, oa, hex_unplace(K), v);
    else
        add_to_members(w, dacc, ob, hex_unplace(K - HEX_SIZE), v);
}

// ------------------------------------------------------------------------------------------------ lane order (LaneSort)
// k_td_play is bound by L1 misses of its table gathers: a CU's 32 KB L1 sees ~4 000 unrelated boards per launch.  Which
// cache lines a board's COLD tuples touch is decided by where its big tiles sit (the small ones come and go every move),
// so lanes with the same big-tile pattern use the same lines.  Every few steps the lanes are therefore re-ordered by that
// pattern, and the next k_td_play reads its lanes through the permutation and writes them back in the new order (no
// separate copy pass).  The host never sees the order: lane_id travels with the lane and every lane-addressed entry point
// of the ABI first restores the identity order (k_restore_order).
//
// The re-order is a counting sort on a 16-bit key — one bit per cell: "the tile is above SORT_TILE" — in three launches
// (round 2 called rocPRIM's radix sort on a 48-bit key: 21 launches, 0.19 ms):
//   k_sort_count   one 1 024-thread workgroup per 4 096 lanes.  All 65 536 counters of the tile live in LDS (16 bits each,
//                  two per word): a lane's ds_add_rtn gives its rank among the tile's lanes with the same key; the first
//                  lane of every key present then reserves the tile's share of the key's bucket with ONE global atomic
//                  (<= 256 atomics per key and sort, however skewed the keys) and passes the reserved offset on to the
//                  others through LDS;
//   k_sort_scan    exclusive scan of the 65 536 bucket sizes (64 independent workgroups);
//   k_sort_scatter perm[bucket start + offset] = position; counters cleared for the next sort.
// Inside a bucket the order is that of the tiles' reservations, i.e. not reproducible from run to run; nothing observable
// depends on it (the lanes' games do not, the fixed-point sums of the update do not).
constexpr uint32_t SORT_TILE = 5;               // default threshold (G2048_SORT_TILE): key bit set for tiles above 2^5
constexpr uint32_t SORT_KEYS = 65536, SORT_TPB = 1024, SORT_PER_THREAD = 4, SORT_LANES_PER_WG = SORT_TPB * SORT_PER_THREAD;

// (round 3, late) WHERE the big tiles sit and WHAT they are: value >> 1 of every tile above thr — what such a tile contributes to
// the address of a table line (a 64-byte line of a table in table_place order holds the 16 entries that differ in bit 0 of
// the four cells) — folded into 16 bits by a multiplicative hash.  Lanes with the same big tiles
// in the same places share a bucket; the order of the buckets means nothing.
constexpr uint32_t SORT_BY_VALUE = 16u;         // flag in the `thr` argument of the sort kernels
__device__ __forceinline__ uint32_t sort_key_values(const Board& b, uint32_t thr) {
    uint32_t h = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t v = (b.r[r] >> (8 * c)) & 0xFFu;
            h = h * 0x9E3779B1u + (v > thr ? (v >> 1) & 7u : 0u);
        }
    return (h >> 16) ^ (h & 0xFFFFu);
}

// one bit per cell, row 0 in the top nibble: tile > thr (tiles are < 128, so the carry into bit 7 of each byte lane is the test)
__device__ __forceinline__ uint32_t sort_key(const Board& b, uint32_t thr) {
    if (thr & SORT_BY_VALUE) return sort_key_values(b, thr & 15u);
    const uint32_t bias = (0x7Fu - thr) * 0x01010101u;
    uint32_t key = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t m = ((b.r[r] + bias) & 0x80808080u) >> 7;          // bits 0, 8, 16, 24
        key = (key << 4) | ((m | m >> 7 | m >> 14 | m >> 21) & 0xFu);
    }
    return key;
}

__global__ __launch_bounds__(WG) void k_iota(uint32_t* v, uint32_t n) {
    uint32_t i = blockIdx.x * WG + threadIdx.x;
    if (i < n) v[i] = i;
}

__global__ __launch_bounds__(SORT_TPB) void k_sort_count(const uint4* boards, uint32_t B, uint32_t thr, uint32_t* cnt, uint16_t* key16, uint32_t* off) {
    __shared__ uint32_t tab[SORT_KEYS / 2];                 // per key: lanes of this tile (16 bits), later the tile-local index of the key's first lane
    __shared__ uint32_t reserved[SORT_LANES_PER_WG];        // [tile-local index of a key's first lane] = start of the tile's share of the bucket
    for (uint32_t j = threadIdx.x; j < SORT_KEYS / 2; j += SORT_TPB) tab[j] = 0;
    __syncthreads();
    uint32_t key[SORT_PER_THREAD], rank[SORT_PER_THREAD];
    const uint32_t i0 = blockIdx.x * SORT_LANES_PER_WG + threadIdx.x;
#pragma unroll
    for (uint32_t u = 0; u < SORT_PER_THREAD; ++u) {
        const uint32_t i = i0 + u * SORT_TPB;
        key[u] = 0;
        rank[u] = 0;
        if (i < B) {
            // (one returning LDS atomic per lane.  Folding the lanes of a wave that share a key into one atomic first — a
            // readlane loop over the wave's distinct keys — was measured slower, 51 against 28 us: mid-game boards put
            // dozens of different keys into a wave)
            key[u] = sort_key(ld_board(boards, i), thr);
            const uint32_t sh = 16u * (key[u] & 1u);
            rank[u] = (atomicAdd(&tab[key[u] >> 1], 1u << sh) >> sh) & 0xFFFFu;      // (<= 4 096 per half: no carry between the halves)
        }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t u = 0; u < SORT_PER_THREAD; ++u)
        if (i0 + u * SORT_TPB < B && rank[u] == 0) {
            const uint32_t total = (tab[key[u] >> 1] >> (16u * (key[u] & 1u))) & 0xFFFFu;
            reserved[u * SORT_TPB + threadIdx.x] = atomicAdd(&cnt[key[u]], total);
        }
    __syncthreads();
    uint16_t* const first = reinterpret_cast<uint16_t*>(tab);
#pragma unroll
    for (uint32_t u = 0; u < SORT_PER_THREAD; ++u)
        if (i0 + u * SORT_TPB < B && rank[u] == 0) first[key[u]] = (uint16_t)(u * SORT_TPB + threadIdx.x);
    __syncthreads();
#pragma unroll
    for (uint32_t u = 0; u < SORT_PER_THREAD; ++u) {
        const uint32_t i = i0 + u * SORT_TPB;
        if (i < B) {
            key16[i] = (uint16_t)key[u];
            off[i] = reserved[first[key[u]]] + rank[u];
        }
    }
}

// start[k] = number of lanes with a smaller key.  64 workgroups, one per 1 024 buckets, none waiting for another: each first adds
// up everything in front of its segment (the whole counter array is 256 KB and sits in L2: at most 63 coalesced loads per
// thread), then scans its own 1 024 counters.  (Round 3's first form was one workgroup walking all 65 536 buckets: 73 us on
// the side stream, 3.4 % of all kernel time for 1/256 of the chip.)  The counters are cleared by k_sort_scatter, which runs
// behind this kernel on the same stream — here every workgroup still reads the others' segments.
constexpr uint32_t SCAN_SEG = SORT_TPB;
__global__ __launch_bounds__(SORT_TPB) void k_sort_scan(const uint32_t* cnt, uint32_t* start) {
    __shared__ uint32_t wave_sum[SORT_TPB / 64], wave_front[SORT_TPB / 64];
    const uint32_t seg = blockIdx.x * SCAN_SEG, lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    uint32_t front = 0;                                     // this thread's share of the buckets in front of the segment
    for (uint32_t i = threadIdx.x; i < seg; i += SORT_TPB) front += cnt[i];
    const uint32_t mine = cnt[seg + threadIdx.x];
    uint32_t incl = mine;                                   // inclusive scan inside the wave
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d, 64);
        if (lane >= d) incl += up;
    }
#pragma unroll
    for (uint32_t d = 32; d > 0; d >>= 1) front += __shfl_down(front, d, 64);
    if (lane == 63u) wave_sum[wv] = incl;
    if (lane == 0u) wave_front[wv] = front;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t k = 0; k < SORT_TPB / 64; ++k) base += wave_front[k] + (k < wv ? wave_sum[k] : 0u);
    start[seg + threadIdx.x] = base + incl - mine;
}

__global__ __launch_bounds__(WG) void k_sort_scatter(const uint16_t* key16, const uint32_t* off, const uint32_t* start, uint32_t B, uint32_t* perm, uint32_t* cnt) {
    const uint32_t i = blockIdx.x * WG + threadIdx.x;
    if (i < B) perm[start[key16[i]] + off[i]] = i;
    if (i < SORT_KEYS) cnt[i] = 0;                          // (the bucket counters, for the next sort: k_sort_count adds into them)
}

struct CarrySet {       // the `state` of QAgent.episode and its orbit indices (prev[cur], oidx[cur])
    uint4* prev;
    uint8_t* oidx;
};

// position j holds lane lane_id[j]: put everything back where the host expects it
template <bool ORBITS>
__global__ __launch_bounds__(WG) void k_restore_order(LaneSet in, LaneSet out, CarrySet cin, CarrySet cout, uint32_t B) {
    uint32_t j = blockIdx.x * WG + threadIdx.x;
    if (j >= B) return;
    const uint32_t id = in.lane_id[j];
    out.boards[id] = in.boards[j];
    out.scores[id] = in.scores[j];
    out.rng[id] = in.rng[j];
    out.label[id] = in.label[j];
    out.flags[id] = in.flags[j];
    out.last_move[id] = in.last_move[j];
    out.lane_id[id] = id;
    cout.prev[id] = cin.prev[j];
    if (ORBITS) {
        const OrbitIdx a = orbit_idx(cin.oidx, B), b = orbit_idx(cout.oidx, B);
#pragma unroll
        for (int v = 0; v < 4; ++v) b.q[(size_t)v * B + id] = a.q[(size_t)v * B + j];
        b.x[id] = a.x[j];
        b.c[id] = a.c[j];
    }
}

// init_weights (r_learning.py:139-149): U[0, scale) per slot, counter-based so any rank can build the same table
// (`placed`: the leading slots that live at table_place())
__global__ __launch_bounds__(WG) void k_weights_init(float* w, uint64_t count, uint64_t placed, uint64_t seed, float scale) {
    uint64_t i = (uint64_t)blockIdx.x * WG + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * WG;
    for (; i < count; i += stride) {
        uint64_t x = seed + i;
        uint64_t z = splitmix64(x);
        w[i < placed ? table_place_any((uint32_t)i) : i] = (float)(z >> 40) * (1.0f / 16777216.0f) * scale;
    }
}

__global__ __launch_bounds__(WG) void k_delta_sub(const float* w, const float* w0, float* delta, uint64_t count) {
    uint64_t i = (uint64_t)blockIdx.x * WG + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * WG;
    for (; i < count; i += stride) delta[i] = w[i] - w0[i];
}

// W = W0 + (delta summed over the ranks); the next epoch starts here: W0 = W, accumulator cleared
// In the kernels below the table-side arrays (w, w0, dacc) are in memory order and the exchange buffer is in INDEX order
// for its first `placed` entries (a caller's buffer: the ABI speaks the reference's order) — placed = 0 for the context's
// own buffers, which only ever meet other memory-ordered arrays.
__global__ __launch_bounds__(WG) void k_delta_out(const float* dacc, float* dst, uint64_t count, uint64_t placed) {
    uint64_t i = (uint64_t)blockIdx.x * WG + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * WG;
    for (; i < count; i += stride) dst[i] = dacc[i < placed ? table_place_any((uint32_t)i) : i];
}

// the whole table between index order (`flat`, a staging buffer) and memory order (g2048_weights_get / _set)
__global__ __launch_bounds__(WG) void k_table_in(float* w, const float* flat, uint64_t count) {
    uint64_t i = (uint64_t)blockIdx.x * WG + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * WG;
    for (; i < count; i += stride) w[table_place_any((uint32_t)i)] = flat[i];
}

__global__ __launch_bounds__(WG) void k_delta_add(float* w, float* w0, const float* delta, float* dacc, uint64_t count, uint64_t placed) {
    uint64_t i = (uint64_t)blockIdx.x * WG + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * WG;
    for (; i < count; i += stride) {
        const uint64_t p = i < placed ? table_place_any((uint32_t)i) : i;
        float v = w0[p] + delta[i];
        w[p] = v;
        w0[p] = v;
        if (dacc) dacc[p] = 0.0f;
    }
}

// Per-slot mean rule across ranks: pack = [delta | touched] with touched = 1 where this rank moved the slot in the epoch;
// after the sum all-reduce the slot moves by the mean over the ranks that touched it.
__global__ __launch_bounds__(WG) void k_delta_pack_touched(const float* dacc, float* pack, uint64_t count, uint64_t placed) {
    uint64_t i = (uint64_t)blockIdx.x * WG + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * WG;
    for (; i < count; i += stride) {
        const float d = dacc[i < placed ? table_place_any((uint32_t)i) : i];
        pack[i] = d;
        pack[count + i] = d != 0.0f ? 1.0f : 0.0f;
    }
}

__global__ __launch_bounds__(WG) void k_delta_add_mean(float* w, float* w0, const float* pack, float* dacc, uint64_t count, uint64_t placed) {
    uint64_t i = (uint64_t)blockIdx.x * WG + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * WG;
    for (; i < count; i += stride) {
        const uint64_t p = i < placed ? table_place_any((uint32_t)i) : i;
        const float n = pack[count + i];
        float v = w0[p] + (n > 1.0f ? pack[i] / n : pack[i]);
        w[p] = v;
        w0[p] = v;
        if (dacc) dacc[p] = 0.0f;
    }
}

}  // namespace

constexpr size_t QCOUNT_WORDS = 32 + 2 * PLAY_SEGS * PLAY_SEG_STRIDE;

// ================================================================================================ host side / C ABI

struct g2048_ctx {
    int device = 0;
    uint32_t B = 0;
    int n = 0, F = 0;
    uint64_t slots = 0, seed = 0, lane0 = 0;
    uint64_t placed = 0;                // the leading table slots that live at table_place_any(): all of them for n >= 4, none for n = 2, 3
    int auto_reset = 1;
    int cur = 0;                        // which half of `prev` holds the current `state`
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_plan = nullptr;
    bool plan_pending = false;          // stats_readback asked for the statistics, replan has not looked at them yet
    uint32_t mirror_seq = 0;            // sequence number of the last apply kernel launched with a mirror
    uint32_t mirror_want = 0;           // the one replan waits for
    uint4* boards = nullptr;
    int32_t* scores = nullptr;
    ulonglong2* rng = nullptr;
    uint4* prev[2] = {nullptr, nullptr};
    uint8_t* oidx[2] = {nullptr, nullptr};      // orbit indices of prev[] (n >= 4)
    uint64_t* wg_clock = nullptr;               // [MAX_SLICES][2] start / end clock of every owner workgroup (last launch)
    uint8_t* statbuf = nullptr;                 // one allocation: hit counters | workgroup clocks (one copy reads both back)
    uint8_t* h_stat = nullptr;                  // pinned host mirrors: statbuf, and the staging of the slices
    void* h_stat_dev = nullptr;                 // h_stat as the device sees it (the apply kernels store the statistics there)
    Slice* h_slices = nullptr;
    std::vector<uint32_t> hits_seen;            // the counters are cumulative (mod 2^32); what the last readback saw
    std::vector<Slice> plan;                    // host copy of the slices in use
    float* label = nullptr;
    uint8_t* flags = nullptr;
    float* dw1 = nullptr;               // main record of every lane (0 = none)
    uint4* qstate = nullptr;            // terminal-record queue
    float* qdw = nullptr;
    uint32_t* qcount = nullptr;         // [QCOUNT_WORDS]: this / next step's queue length [0,1], largest |dw| bits [2,3], from [32] on k_td_play's block counters (two sets)
    uint16_t* last_move = nullptr;      // what every lane did in the latest TD step (g2048_get_last_move)
    // lane order (LaneSort): boards / scores / rng / label / flags / last_move / lane_id above are the CURRENT set; `alt` is
    // the other one (allocated with the first re-ordering)
    uint32_t* lane_id = nullptr;
    LaneSet alt = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool permuted = false;              // the lanes are not in identity order
    uint32_t sort_every = 0;            // re-order the lanes every this many TD steps (0 = never)
    uint32_t steps_since_sort = 0;
    // the sort itself runs on a side stream beside the following steps' kernels; its permutation is applied `sort_lag` steps
    // after the boards it was computed from (lanes keep their positions in between, so a stale order is still a valid one)
    hipStream_t sort_stream = nullptr;
    hipEvent_t ev_sort_go = nullptr, ev_sort_done = nullptr;
    bool sort_pending = false;          // sort_perm is being (or has been) computed and waits to be applied
    bool sort_issued = false;           // ev_sort_done has been recorded at least once
    uint32_t sort_wait = 0;             // steps until the pending permutation is applied
    uint16_t* sort_key16 = nullptr;     // [B] the lanes' keys
    uint32_t *sort_off = nullptr, *sort_perm = nullptr;     // [B] offset inside the key's bucket; the permutation
    uint32_t *sort_cnt = nullptr, *sort_start = nullptr;    // [SORT_KEYS] bucket sizes (zero between sorts), bucket starts
    GameLog log = {0, 0, nullptr, nullptr, nullptr, nullptr};
    uint32_t step_parity = 0;
    float *w = nullptr, *w0 = nullptr, *delta = nullptr;
    // multi-GPU epoch state (g2048_delta_begin): W0 = table at the start of the epoch; `delta` accumulates every add this
    // context makes to the table (null until tracking is on); `pack` = all-reduce buffer ([delta | touched] for the mean rule)
    bool tracking = false;
    float* pack = nullptr;
    size_t pack_count = 0;
    uint32_t hex_young = 0;             // TD steps left with the exact layout of the f_6 bins (k_hex_count: young, synchronised boards)
    HexBufs hex = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0};   // n = 6: binned update of the f_6 orbits (k_hex_*)
    void* comm = nullptr;               // ncclComm_t of g2048_comm_init
    int comm_rank = 0, comm_ranks = 1;
    Stats* stats = nullptr;
    Slice* slices = nullptr;            // LDS-owner update plan (device copy, capacity MAX_SLICES)
    uint32_t n_slices = 0;
    uint32_t* hits = nullptr;           // adds per table chunk since the last re-plan (load statistics)
    uint32_t n_chunks = 0;
    unsigned play_wgs = 0;          // persistent grid of k_td_play (0 = not determined yet)
    uint32_t owned_total = 0;       // D[0 .. owned_total): the LDS-owned orbit tables (cleared by the host after k_apply_orbits)
    uint32_t steps_since_plan = 0;  // launches of the owner kernel under the current plan
    uint32_t steps_since_read = 0;  // ... since the hit counters were last read back
    uint32_t replan_interval = 1;   // steps until the next unconditional replan: 1, 2, 4, ... replan_every after a (re)start
    double makespan_ref = 0;        // owner kernel makespan (100 MHz ticks) of the first launch under the current plan
    bool plan_measured = false;     // the load is a measurement (not the creation-time prior)
    // QuadOrder.  The four-cell orbits' tables start "hot first" (features.hpp, quad_place: the indices with every cell <= 10 in
    // chunk 0, which then holds every add of an agent that makes no 2 048s: one scan per orbit instead of 2 - 3).  Once boards
    // with bigger tiles send more than a few per cent of an orbit's adds elsewhere, the four chunks behind chunk 0 would all
    // be worth holding — five scans where the plain index order needs four — so the context then switches to the plain order
    // (every index in [QUAD_HOT, QUAD_HOT + 65 536), chunk 0 unused) until g2048_reset.  D is empty between two steps, which is
    // when replan() switches.  G2048_QUAD_ORDER = hot | plain pins it.
    bool quad_plain = false;
    bool discard_stats = false;     // the hit counters not read yet were counted before a g2048_reset changed the order back: skip them once
    hipEvent_t ev_table = nullptr;  // last table-touching launch on `stream` (contexts that share a table wait on it)
    g2048_ctx* parent = nullptr;    // owner of the shared table (g2048_create_shared); null: this context owns `w`
    uint32_t shared_users = 0;      // (owner only) contexts created on this table with g2048_create_shared and still alive
    g2048_ctx* last_user = nullptr; // (owner only) whose stream carried the latest launch that reads or writes the table
    // experiment knobs (environment, read ONCE at creation): defaults are the measured best
    struct Knobs {
        uint32_t replan_every = 8;      // upper end of the replan schedule: the board distribution drifts with the games' age
        double imbalance = 1.15;        // replan as soon as the owner kernel's makespan exceeds this multiple of makespan_ref
        int feedback_each_step = 1;     // read the workgroup clocks back after every step (big batches only)
        double add_cost = 3.0, thr = 0.01, fixed_ratio = 0.25;
        int plan_feedback = 1, plan_xcd = 1, debug_plan = 0;
        double plan_mixed = 6.0;        // XCD-resident plan: chunks that deserve fewer workgroups than this are scanned flat (0: none)
        int quad_order = 0;             // 0: hot first, switching to the plain order by the measured shares; 1: always hot first; 2: always plain
        double quad_switch = 0.96;      // switch when chunk 0 holds less than this share of the four-cell orbits' adds
        double plan_fixed_us = 3.0;     // the part of a workgroup's time that does not shrink with its share of the records (planner's cost model)
        unsigned play_wgs = 0;
        uint32_t play_dynamic = 2;      // full rounds of k_td_play's lane blocks left to the counter (besides the last, partial one)
        uint32_t sort_every = 8;        // default of g2048_set_lane_sort for new contexts (G2048_SORT_EVERY); 0 = never
        uint32_t sort_min_batch = 1u << 17;     // smaller batches keep their lane order
        uint32_t sort_tile = SORT_TILE; // key bit of a cell: tile above 2^this (G2048_SORT_TILE)
        uint32_t plan_poll = 1;         // the planner's statistics are announced by the apply kernel itself (0: an event behind it; G2048_PLAN_POLL)
        uint32_t sort_values = 1;       // the key also tells the big tiles' values apart (hashed; G2048_SORT_VALUES=0: positions only, round 2's key)
        uint32_t sort_lag = 2;          // steps between the boards a sort looks at and the step that applies it (G2048_SORT_LAG; 0 = sort in line)
        int hex_bins = 1;               // n = 6: f_6 orbits through k_hex_* (0: k_td_update_tail's scattered atomics)
        int comm_rsag = 0;              // g2048_allreduce_deltas as ncclReduceScatter + ncclAllGather (G2048_COMM_ALGO=rsag)
        int delta_accum = 0;            // multi-GPU epoch delta: 0 = W - W0 when it is asked for, 1 = every add of a step mirrored in an accumulator
        int mean_one_pass = 1;          // per-slot mean rule: counts packed beside the sums (0: always two accumulation passes)
        int play_hot = 1;               // k_td_play reads the first entries of every four-cell table (memory order) from an LDS copy (n >= 4, big batches): -8 % per step
        uint32_t play_hot_min = 1u << 18;   // smallest batch that takes that path (a workgroup copies 136 KB per launch: break-even at 2^17 lanes)
    } knob;
    std::vector<double> load;           // smoothed adds per step per chunk
    std::vector<double> work;           // measured workgroup time x workgroups per chunk (clock ticks; 0 = not measured yet)
    float* D = nullptr;                 // per-orbit delta tables (n >= 4; mean rule: also the sums for n = 2, 3)
    float* Dcnt = nullptr;              // mean rule: how many adds each slot of D received
    float* D2 = nullptr;                // second buffer of D[0 .. owned_total) and of Dcnt (n >= 4): see k_apply_orbits
    float* Dcnt2 = nullptr;
    uint32_t dpar = 0;                  // which buffer this step's LDS-owner sums go to
    int update_rule = 0;                // 0: add every dw (QAgent.update), 1: per-slot mean
    OrbitTable orbits = {};
    int update_mode = 1;                // 1: LDS-owner update (default), 0: global fp32 atomics
    bool owns_table = true;             // false: `w` belongs to the parent context (g2048_create_shared)
    void* scratch = nullptr;
    size_t scratch_bytes = 0;
    void* la_ws = nullptr;              // workspace of the look-ahead trees (lookahead.hip), grown on demand
    size_t la_ws_bytes = 0;
    std::string err;
};

namespace {

int fail(g2048_ctx* c, int code, const char* what, hipError_t e = hipSuccess) {
    if (c) {
        char buf[512];
        if (e != hipSuccess)
            snprintf(buf, sizeof buf, "%s: %s (%s)", what, hipGetErrorString(e), hipGetErrorName(e));
        else
            snprintf(buf, sizeof buf, "%s", what);
        c->err = buf;
    }
    return code;
}

#define HIP_TRY(c, call)                                                     \
    do {                                                                     \
        hipError_t e_ = (call);                                              \
        if (e_ != hipSuccess) return fail((c), G2048_ERR_HIP, #call, e_);    \
    } while (0)

#define NEED(c, cond, msg) \
    do {                   \
        if (!(cond)) return fail((c), G2048_ERR_ARG, msg); \
    } while (0)

#define NEED_TABLE(c) \
    do {              \
        if ((c)->n == 0) return fail((c), G2048_ERR_STATE, "context has no weight table (n_tuple == 0)"); \
    } while (0)

inline unsigned grid_for(uint64_t threads) { return (unsigned)((threads + WG - 1) / WG); }

// Contexts that share a weight table (g2048_create_shared) run on their own streams.  The caller serialises its CALLS on
// such contexts; these two order the DEVICE work behind them: before a context's launches read or write the table its
// stream waits for the latest launch any other context made on it (an event on the owner, re-recorded after every use).
// With no sharing this is free.
g2048_ctx* table_root(g2048_ctx* c) { return c->parent ? c->parent : c; }
struct TableUse {
    g2048_ctx* c;
    int rc = G2048_OK;
    explicit TableUse(g2048_ctx* ctx) : c(ctx) {
        g2048_ctx* r = table_root(c);
        if (r->shared_users && r->last_user && r->last_user != c && hipStreamWaitEvent(c->stream, r->ev_table, 0) != hipSuccess) {
            c->err = "hipStreamWaitEvent(table)";
            rc = G2048_ERR_HIP;
        }
    }
    ~TableUse() {
        g2048_ctx* r = table_root(c);
        if (rc == G2048_OK && r->shared_users && hipEventRecord(r->ev_table, c->stream) == hipSuccess) r->last_user = c;
    }
};
#define USE_TABLE(c)            \
    TableUse table_use_(c);     \
    if (table_use_.rc) return table_use_.rc

// experiment knobs: the environment is read once, when a context is created
void read_knobs(g2048_ctx* c) {
    g2048_ctx::Knobs& k = c->knob;
    if (const char* e = getenv("G2048_REPLAN_EVERY")) k.replan_every = (uint32_t)std::max(1, atoi(e));
    if (const char* e = getenv("G2048_PLAN_IMBALANCE")) k.imbalance = atof(e);
    if (const char* e = getenv("G2048_PLAN_EACH_STEP")) k.feedback_each_step = atoi(e);
    if (const char* e = getenv("G2048_PLAN_FIXEDRATIO")) k.fixed_ratio = atof(e);
    if (const char* e = getenv("G2048_PLAN_ADDCOST")) k.add_cost = atof(e);
    if (const char* e = getenv("G2048_PLAN_THR")) k.thr = atof(e);
    if (const char* e = getenv("G2048_PLAN_FEEDBACK")) k.plan_feedback = atoi(e);
    if (const char* e = getenv("G2048_PLAN_XCD")) k.plan_xcd = atoi(e);
    if (const char* e = getenv("G2048_PLAN_MIXED")) k.plan_mixed = atof(e);
    if (const char* e = getenv("G2048_PLAN_FIXED_US")) k.plan_fixed_us = atof(e);
    if (const char* e = getenv("G2048_QUAD_ORDER")) k.quad_order = !strcmp(e, "hot") ? 1 : !strcmp(e, "plain") ? 2 : 0;
    if (const char* e = getenv("G2048_QUAD_SWITCH")) k.quad_switch = atof(e);
    if (getenv("G2048_DEBUG_PLAN")) k.debug_plan = 1;
    if (const char* e = getenv("G2048_PLAY_WGS")) k.play_wgs = (unsigned)atoi(e);
    if (const char* e = getenv("G2048_PLAY_DYNAMIC")) k.play_dynamic = (uint32_t)atoi(e);
    if (const char* e = getenv("G2048_PLAY_HOT")) k.play_hot = atoi(e);
    if (const char* e = getenv("G2048_MEAN_ONE_PASS")) k.mean_one_pass = atoi(e);
    if (const char* e = getenv("G2048_HEX_BINS")) k.hex_bins = atoi(e);
    if (const char* e = getenv("G2048_COMM_ALGO")) k.comm_rsag = strcmp(e, "rsag") == 0;
    if (const char* e = getenv("G2048_SORT_EVERY")) k.sort_every = (uint32_t)atoi(e);
    if (const char* e = getenv("G2048_SORT_MIN")) k.sort_min_batch = (uint32_t)atoi(e);
    if (const char* e = getenv("G2048_SORT_LAG")) k.sort_lag = (uint32_t)atoi(e);
    if (const char* e = getenv("G2048_PLAN_POLL")) k.plan_poll = (uint32_t)atoi(e);
    if (const char* e = getenv("G2048_SORT_VALUES")) k.sort_values = (uint32_t)atoi(e);
    if (const char* e = getenv("G2048_SORT_TILE")) k.sort_tile = (uint32_t)atoi(e) < 15u ? (uint32_t)atoi(e) : 15u;
    if (const char* e = getenv("G2048_DELTA_ACCUM")) k.delta_accum = atoi(e);
    if (const char* e = getenv("G2048_PLAY_HOT_MIN")) k.play_hot_min = (uint32_t)atoi(e);
}

int bind(g2048_ctx* c) {
    HIP_TRY(c, hipSetDevice(c->device));
    return G2048_OK;
}

int ensure_scratch(g2048_ctx* c, size_t bytes) {
    if (bytes <= c->scratch_bytes) return G2048_OK;
    if (c->scratch) HIP_TRY(c, hipFree(c->scratch));
    c->scratch = nullptr;
    c->scratch_bytes = 0;
    hipError_t e = hipMalloc(&c->scratch, bytes);
    if (e != hipSuccess) return fail(c, G2048_ERR_NOMEM, "hipMalloc(scratch)", e);
    c->scratch_bytes = bytes;
    return G2048_OK;
}

int h2d(g2048_ctx* c, void* dst, const void* src, size_t bytes) {
    HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return G2048_OK;
}

int d2h(g2048_ctx* c, void* dst, const void* src, size_t bytes) {
    HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return G2048_OK;
}

int launched(g2048_ctx* c, const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(c, G2048_ERR_HIP, what, e);
    return G2048_OK;
}

template <class T>
int dalloc(g2048_ctx* c, T** p, size_t count) {
    hipError_t e = hipMalloc((void**)p, count * sizeof(T));
    if (e != hipSuccess) return fail(c, G2048_ERR_NOMEM, "hipMalloc", e);
    return G2048_OK;
}

// ---- lane order (LaneSort): see k_sort_count
LaneSet current_set(g2048_ctx* c) { return LaneSet{c->boards, c->scores, c->rng, c->label, c->flags, c->lane_id, c->last_move}; }

void adopt_set(g2048_ctx* c, const LaneSet& s) {
    c->boards = s.boards; c->scores = s.scores; c->rng = s.rng; c->label = s.label; c->flags = s.flags; c->lane_id = s.lane_id; c->last_move = s.last_move;
}

int lane_sort_prepare(g2048_ctx* c) {
    if (c->alt.boards) return G2048_OK;
    const size_t B = c->B;
    int rc;
    if ((rc = dalloc(c, &c->alt.boards, B)) || (rc = dalloc(c, &c->alt.scores, B)) || (rc = dalloc(c, &c->alt.rng, B)) ||
        (rc = dalloc(c, &c->alt.label, B)) || (rc = dalloc(c, &c->alt.flags, B)) || (rc = dalloc(c, &c->alt.lane_id, B)) ||
        (rc = dalloc(c, &c->alt.last_move, B)) || (rc = dalloc(c, &c->sort_key16, B)) || (rc = dalloc(c, &c->sort_off, B)) ||
        (rc = dalloc(c, &c->sort_perm, B)) || (rc = dalloc(c, &c->sort_cnt, SORT_KEYS)) || (rc = dalloc(c, &c->sort_start, SORT_KEYS)))
        return rc;
    HIP_TRY(c, hipMemsetAsync(c->alt.last_move, 0, B * 2, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->sort_cnt, 0, SORT_KEYS * sizeof(uint32_t), c->stream));
    HIP_TRY(c, hipStreamCreateWithFlags(&c->sort_stream, hipStreamNonBlocking));
    HIP_TRY(c, hipEventCreateWithFlags(&c->ev_sort_go, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&c->ev_sort_done, hipEventDisableTiming));
    return G2048_OK;
}

// the current boards -> sort_perm: position i of the new order takes the lane now at position sort_perm[i].
// `side`: the keys are counted on the context's stream (the boards as they are now); the scan and the scatter run on the
// side stream and ev_sort_done marks their end; otherwise everything is queued in line.
int lane_sort_permutation(g2048_ctx* c, bool side) {
    if (int rc = lane_sort_prepare(c)) return rc;
    if (c->sort_issued) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_sort_done, 0));       // (an abandoned sort may still be reading its buffers)
    k_sort_count<<<(c->B + SORT_LANES_PER_WG - 1) / SORT_LANES_PER_WG, SORT_TPB, 0, c->stream>>>(c->boards, c->B, c->knob.sort_tile | (c->knob.sort_values ? SORT_BY_VALUE : 0u), c->sort_cnt, c->sort_key16,
                                                                                           c->sort_off);
    hipStream_t st = c->stream;
    if (side) {
        HIP_TRY(c, hipEventRecord(c->ev_sort_go, c->stream));
        HIP_TRY(c, hipStreamWaitEvent(c->sort_stream, c->ev_sort_go, 0));
        st = c->sort_stream;
    }
    k_sort_scan<<<SORT_KEYS / SCAN_SEG, SORT_TPB, 0, st>>>(c->sort_cnt, c->sort_start);
    k_sort_scatter<<<grid_for(c->B > SORT_KEYS ? c->B : SORT_KEYS), WG, 0, st>>>(c->sort_key16, c->sort_off, c->sort_start, c->B, c->sort_perm, c->sort_cnt);
    if (side) {
        HIP_TRY(c, hipEventRecord(c->ev_sort_done, c->sort_stream));
        c->sort_issued = true;
    }
    return launched(c, "lane re-order");
}

// every entry point that addresses lanes by index calls this first: back to identity order (a no-op unless a TD step
// re-ordered the lanes since)
int ensure_identity(g2048_ctx* c) {
    if (!c->permuted) return G2048_OK;
    if (int rc = lane_sort_prepare(c)) return rc;
    const CarrySet cin{c->prev[c->cur], c->oidx[c->cur]}, cout{c->prev[c->cur ^ 1], c->oidx[c->cur ^ 1]};
    if (c->n >= 4)
        k_restore_order<true><<<grid_for(c->B), WG, 0, c->stream>>>(current_set(c), c->alt, cin, cout, c->B);
    else
        k_restore_order<false><<<grid_for(c->B), WG, 0, c->stream>>>(current_set(c), c->alt, cin, cout, c->B);
    const LaneSet was = current_set(c);
    adopt_set(c, c->alt);
    c->alt = was;
    c->cur ^= 1;                    // the carried state now lives in the other half of prev / oidx
    c->permuted = false;
    c->steps_since_sort = 0;
    c->sort_pending = false;        // a permutation computed for the old positions must not be applied to the new ones
    return launched(c, "k_restore_order");
}
#define NEED_IDENTITY(c)                     \
    do {                                     \
        if (int rc_ = ensure_identity(c)) return rc_; \
    } while (0)

// dispatch on the n-tuple size
#define BY_N(c, EXPR)                                                      \
    switch ((c)->n) {                                                      \
        case 2: { constexpr int N = 2; EXPR; } break;                      \
        case 3: { constexpr int N = 3; EXPR; } break;                      \
        case 4: { constexpr int N = 4; EXPR; } break;                      \
        case 5: { constexpr int N = 5; EXPR; } break;                      \
        case 6: { constexpr int N = 6; EXPR; } break;                      \
        default: return fail((c), G2048_ERR_STATE, "bad n_tuple");         \
    }

// ---- symmetry orbits, found by brute force at context creation (the index functions are host-callable): feature i
// belongs to the orbit of an earlier feature r if some image g and some nibble permutation make
// f_i(x) == perm(f_r(g.x)) on a set of random boards.
template <int N>
uint32_t host_feature_index(const Packed& p, int f) {
    uint32_t s[Shape<N>::F];
    feature_slots<N>(p, s);
    return s[f] - feature_offset(N, f);
}

template <int N>
int find_orbits(g2048_ctx* c) {
    constexpr int NF = Shape<N>::F;
    constexpr int NB = 24;
    std::vector<Packed> boards;
    uint64_t x = 0x2048;
    for (int t = 0; t < NB; ++t) {
        Board b;
        for (int r = 0; r < 4; ++r) b.r[r] = (uint32_t)(splitmix64(x) & 0x0F0F0F0Fu) % 0x0E0E0E0Fu;   // tiles 0..13 (f_6 clamps at 13)
        for (int r = 0; r < 4; ++r)
            for (int col = 0; col < 4; ++col)
                if (((b.r[r] >> (8 * col)) & 0xFFu) > 13u) b.r[r] &= ~(0xFFu << (8 * col));
        boards.push_back(pack_board(b));
    }
    auto digits_of = [](uint32_t idx, uint32_t radix, uint32_t nd, uint32_t* out) {
        for (uint32_t p = 0; p < nd; ++p) {
            out[p] = idx % radix;
            idx /= radix;
        }
    };
    OrbitTable& T = c->orbits;
    T.count = 0;
    T.total = 0;
    std::vector<int> rep_of;
    for (int i = 0; i < NF; ++i) {
        const uint32_t nd = N < 4 ? (uint32_t)N : i < 17 ? 4u : (i < 21 ? 5u : 6u), radix = (N < 4 || i < 21) ? 16u : 14u;
        uint32_t mine[NB][6];
        for (int t = 0; t < NB; ++t) digits_of(host_feature_index<N>(boards[t], i), radix, nd, mine[t]);
        bool placed = false;
        for (uint32_t o = 0; o < T.count && !placed; ++o) {
            if (T.o[o].digits != nd || T.o[o].radix != radix) continue;
            for (uint32_t g = 0; g < 8 && !placed; ++g) {
                uint32_t theirs[NB][6];
                for (int t = 0; t < NB; ++t) digits_of(host_feature_index<N>(d4_image(boards[t], g), rep_of[o]), radix, nd, theirs[t]);
                // out digit p of feature i must equal some digit q of the representative on every board
                uint32_t perm = 0, used = 0;
                bool ok = true;
                for (uint32_t pp = 0; pp < nd && ok; ++pp) {
                    int found = -1;
                    for (uint32_t q = 0; q < nd && found < 0; ++q) {
                        if ((used >> q) & 1u) continue;
                        bool same = true;
                        for (int t = 0; t < NB && same; ++t) same = mine[t][pp] == theirs[t][q];
                        if (same) found = (int)q;
                    }
                    if (found < 0) {
                        ok = false;
                    } else {
                        used |= 1u << found;
                        perm |= (uint32_t)found << (3u * pp);
                    }
                }
                if (ok) {
                    if (T.o[o].nmem >= MAX_MEMBERS) return fail(c, G2048_ERR_STATE, "orbit larger than expected");
                    T.o[o].off[T.o[o].nmem] = feature_offset(N, i);
                    T.o[o].perm[T.o[o].nmem] = perm;
                    ++T.o[o].nmem;
                    placed = true;
                }
            }
        }
        if (!placed) {
            if (T.count >= MAX_ORBITS) return fail(c, G2048_ERR_STATE, "more orbits than expected");
            OrbitInfo& oi = T.o[T.count++];
            oi = OrbitInfo{};
            oi.base = T.total;
            oi.size = (N >= 4 && nd == 4u) ? QUAD_DSIZE : feature_size(N, i);       // (four-cell orbits: quad_place order, features.hpp)
            oi.digits = nd;
            oi.radix = radix;
            oi.nmem = 1;
            oi.off[0] = feature_offset(N, i);
            for (uint32_t q = 0; q < nd; ++q) oi.perm[0] |= q << (3u * q);
            T.total += oi.size;
            rep_of.push_back(i);
        }
    }
    // stabiliser of every representative, and the check of the kernels' coset masks
    for (uint32_t o = 0; o < T.count; ++o) {
        OrbitInfo& oi = T.o[o];
        oi.nstab = 1;
        oi.sperm[0] = oi.perm[0];                                   // identity
        uint32_t mask;
        if constexpr (N < 4) {
            using SO = SmallOrbits<N < 4 ? N : 2>;
            if ((int)T.count != SO::COUNT || rep_of[o] != SO::rep((int)o) || oi.size != SO::SIZE) return fail(c, G2048_ERR_STATE, "unexpected orbit structure");
            c->owned_total = oi.base + oi.size;
            mask = SO::mask((int)o);
        } else if (oi.radix == 16u) {
            if (o >= 6 || rep_of[o] != ORBIT_REPS[o]) return fail(c, G2048_ERR_STATE, "unexpected orbit structure");
            c->owned_total = oi.base + oi.size;
            mask = COSET_MASK[o];
        } else {
            if (o < 6 || o >= 8) return fail(c, G2048_ERR_STATE, "unexpected f_6 orbit structure");
            mask = HEX_COSET_MASK[o - 6];
        }
        const uint32_t nd = oi.digits, radix = oi.radix;
        uint32_t mine[NB][6];
        for (int t = 0; t < NB; ++t) digits_of(host_feature_index<N>(boards[t], rep_of[o]), radix, nd, mine[t]);
        for (uint32_t g = 1; g < 8; ++g) {
            uint32_t theirs[NB][6];
            for (int t = 0; t < NB; ++t) digits_of(host_feature_index<N>(d4_image(boards[t], g), rep_of[o]), radix, nd, theirs[t]);
            uint32_t perm = 0, used = 0;
            bool ok = true;
            for (uint32_t pp = 0; pp < nd && ok; ++pp) {
                int found = -1;
                for (uint32_t q = 0; q < nd && found < 0; ++q) {
                    if ((used >> q) & 1u) continue;
                    bool same = true;
                    for (int t = 0; t < NB && same; ++t) same = mine[t][pp] == theirs[t][q];
                    if (same) found = (int)q;
                }
                if (found < 0) {
                    ok = false;
                } else {
                    used |= 1u << found;
                    perm |= (uint32_t)found << (3u * pp);
                }
            }
            if (ok) oi.sperm[oi.nstab++] = perm;
        }
        if (radix != 16u && oi.nstab > 2) return fail(c, G2048_ERR_STATE, "unexpected f_6 stabiliser");      // k_apply_orbits pairs k with sigma(k)
        for (int t = 0; t < NB; ++t) {
            std::vector<uint32_t> all, folded;
            for (uint32_t g = 0; g < 8; ++g) {
                const uint32_t idx = host_feature_index<N>(d4_image(boards[t], g), rep_of[o]);
                all.push_back(idx);
                if ((mask >> g) & 1u)
                    for (uint32_t st = 0; st < oi.nstab; ++st) folded.push_back(permute_digits(idx, oi.sperm[st], nd, radix));
            }
            std::sort(all.begin(), all.end());
            std::sort(folded.begin(), folded.end());
            if (all != folded) return fail(c, G2048_ERR_STATE, "coset mask does not reproduce the 8 images");
        }
    }
    if (N == 6) {       // k_td_update_tail hard-codes the two f_6 representatives
        if (T.count != 8 || rep_of[6] != 21 || rep_of[7] != 22) return fail(c, G2048_ERR_STATE, "unexpected f_6 orbit structure");
    }
    return G2048_OK;
}

// ---- LDS-owner plan.  The accumulation space (orbit tables D for n >= 4, the weight table itself for n = 2, 3) is cut
// into chunks of at most OWN_SLOTS slots.  Every chunk is held in LDS by `nparts` workgroups, each scanning a share of
// the records; the ~250 workgroups (one 128 KiB workgroup per CU) are handed out in proportion to each chunk's cost =
// scanning the records + its measured adds per step (hit counters, read back every `replan_every` steps): young
// boards put almost every add into the low half of each table, a trained agent's boards do not.
constexpr uint32_t MAX_SLICES = 1024, HITS_CAP = 128;
constexpr size_t STAT_BYTES = HITS_CAP * 4 + 2 * MAX_SLICES * 8;
static_assert(STAT_BYTES % 8 == 0, "mirror_stats stores {sequence, checksum} as one aligned 8-byte word behind the statistics");
constexpr uint32_t WG_BUDGET = 250;
constexpr uint32_t XCDS = 8, CUS_PER_XCD = 32;

struct ChunkInfo {
    uint32_t variant, tlo, size, dlo;
    double scan;        // relative cost of scanning one record for this chunk
    uint32_t orb_tlo, orb_dlo, chunk0, orb_chunks;      // the orbit table this chunk belongs to (n >= 4)
};

// the per-slot mean rule packs add counts beside the sums, which needs the fixed-point form for every variant
bool chunk_fixed(const g2048_ctx* c, int variant) { return own_fixed(c->n, variant) || c->update_rule == 1; }

std::vector<ChunkInfo> table_chunks(const g2048_ctx* c) {
    std::vector<ChunkInfo> v;
    if (c->n == 2) {            // all four orbit tables in one chunk
        v.push_back({0, 0, c->orbits.total, 0, 3.0, 0, 0, 0, 1});
    } else if (c->n == 3) {     // four orbit tables (4 x 4 096 fixed-point slots = 128 KiB) per chunk
        const uint32_t csz = SmallOrbits<3>::PER_CHUNK * SmallOrbits<3>::SIZE;
        for (uint32_t g = 0; g * csz < c->orbits.total; ++g) v.push_back({g, g * csz, csz, g * csz, 3.0, 0, 0, g, 1});
    } else if (c->n >= 4) {
        for (uint32_t o = 0; o < c->orbits.count; ++o) {
            const OrbitInfo& oi = c->orbits.o[o];
            uint32_t rep = 0;
            if (oi.radix != 16u) continue;          // the f_6 orbits are not LDS-owned (k_td_update_tail)
            for (int j = 0; j < 21 && j < c->F; ++j)
                if (feature_offset(c->n, j) == oi.off[0]) rep = (uint32_t)j;
            const uint32_t first = (uint32_t)v.size();
            if (o >= 6 || (int)rep != ORBIT_REPS[o]) return {};      // (checked by the caller: empty plan = unexpected orbit structure)
            const uint32_t csz = chunk_fixed(c, (int)o) ? FIXED_SLOTS : OWN_SLOTS;
            for (uint32_t lo = 0; lo < oi.size; lo += csz)
                v.push_back({o, oi.off[0] + lo, csz, oi.base + lo, oi.digits == 4 ? 1.0 : 2.0, oi.off[0], oi.base, first, oi.size / csz});
        }
    }
    return v;
}

int build_slices(g2048_ctx* c) {
    if (c->n == 0) return G2048_OK;
    if (!c->slices) {       // first call: orbits, device buffers, and a prior for the load
        {
            int rc = c->n == 2 ? find_orbits<2>(c) : c->n == 3 ? find_orbits<3>(c) : c->n == 4 ? find_orbits<4>(c) : c->n == 5 ? find_orbits<5>(c) : find_orbits<6>(c);
            if (rc) return rc;
            if ((rc = dalloc(c, &c->D, c->orbits.total))) return rc;
            HIP_TRY(c, hipMemset(c->D, 0, (size_t)c->orbits.total * 4));
            if ((rc = dalloc(c, &c->D2, c->owned_total))) return rc;
            HIP_TRY(c, hipMemset(c->D2, 0, (size_t)c->owned_total * 4));
            if (c->n == 6 && c->B <= (1u << 23)) {       // binned update of the f_6 orbits: 16 pairs per record, main + terminal
                c->hex.pairs_cap = (uint32_t)std::min<uint64_t>((uint64_t)2 * HEX_PAIRS * c->B + 2048ull * HEX_BINS, 0xFFFFFFFFull);
                if ((rc = dalloc(c, &c->hex.count, HEX_BINS)) || (rc = dalloc(c, &c->hex.base, HEX_BINS + 1)) ||
                    (rc = dalloc(c, &c->hex.cursor, HEX_BINS)) || (rc = dalloc(c, &c->hex.pairs, (size_t)c->hex.pairs_cap)) ||
                    (rc = dalloc(c, &c->hex.work, HEX_MAX_WORK)) || (rc = dalloc(c, &c->hex.nwork, 1)) || (rc = dalloc(c, &c->hex.touched, HEX_BINS)))
                    return rc;
                k_hex_layout_init<<<1, OWN_WG, 0, c->stream>>>(c->hex);
                c->hex_young = HEX_YOUNG_STEPS;
            }
        }
        if (int rc = dalloc(c, &c->slices, MAX_SLICES)) return rc;
        if (int rc = dalloc(c, &c->statbuf, STAT_BYTES + 64)) return rc;       // (+ mirror_stats' block counter)
        HIP_TRY(c, hipMemset(c->statbuf, 0, STAT_BYTES + 64));
        c->hits = reinterpret_cast<uint32_t*>(c->statbuf);
        c->wg_clock = reinterpret_cast<uint64_t*>(c->statbuf + HITS_CAP * 4);
        HIP_TRY(c, hipHostMalloc((void**)&c->h_stat, STAT_BYTES + 64, hipHostMallocMapped));        // (+ the mirror's sequence number)
        memset(c->h_stat, 0, STAT_BYTES + 64);
        HIP_TRY(c, hipHostGetDevicePointer(&c->h_stat_dev, c->h_stat, 0));
        HIP_TRY(c, hipHostMalloc((void**)&c->h_slices, MAX_SLICES * sizeof(Slice), hipHostMallocDefault));
    }
    const std::vector<ChunkInfo> chunks = table_chunks(c);
    const size_t nc = chunks.size();
    if (nc == 0) return fail(c, G2048_ERR_STATE, "unexpected orbit structure");
    if (nc > HITS_CAP) return fail(c, G2048_ERR_STATE, "more chunks than hit counters");
    if (!c->n_chunks && c->knob.debug_plan)
        for (uint32_t o = 0; o < c->orbits.count; ++o) {
            const OrbitInfo& oi = c->orbits.o[o];
            fprintf(stderr, "[g2048 orbit %u] base %u size %u members:", o, oi.base, oi.size);
            for (uint32_t m = 0; m < oi.nmem; ++m) fprintf(stderr, " (slot %u perm %06o)", oi.off[m], oi.perm[m]);
            fprintf(stderr, "\n");
        }
    if (!c->n_chunks) {
        c->n_chunks = (uint32_t)nc;
        c->hits_seen.assign(nc, 0u);
        c->load.assign(nc, 0.0);
        for (size_t k = 0; k < nc; ++k) {       // fresh games only touch small tiles: the low chunk of every table
            double share = 1.0;
            if (c->n >= 4) {
                const uint32_t rel = (chunks[k].dlo - chunks[k].orb_dlo) / chunks[k].size;
                const uint32_t per16 = 65536u / chunks[k].size;       // chunks per value of the cross's leading nibble
                const double quad_share = c->quad_plain ? (rel == 0 ? 0.0 : rel == 1 ? 0.7 : rel == 2 ? 0.28 : 0.01) : (rel == 0 ? 0.98 : 0.005);
                share = chunks[k].scan == 1.0 ? quad_share : (rel % per16 == 0 && rel < 8 * per16 ? 0.12 : 0.001);
            }
            c->load[k] = 8.0 * c->B * share * (c->n == 2 ? 3 : c->n == 3 ? 3.25 : 1);      // (n = 2, 3: 24 / 26 adds per record and chunk)
        }
    }
    const double add_cost = c->knob.add_cost, thr = c->knob.thr, fixed_ratio = c->knob.fixed_ratio;
    const double B = c->B;
    // which chunks get LDS workgroups: those with at least `thr` of their orbit's adds, and each orbit's busiest
    std::vector<char> in_lds(nc, 1);
    std::vector<uint64_t> duty(nc, 0);      // fallback mask carried by a chunk
    if (c->n >= 4 && c->B >= 4096)
        for (size_t k0 = 0; k0 < nc; k0 += chunks[k0].orb_chunks) {
            const size_t cnt = chunks[k0].orb_chunks;
            double orbit_total = 0;
            size_t best = k0;
            if (c->quad_plain && chunks[k0].scan == 1.0 && cnt > 1) best = k0 + 1;
            for (size_t j = k0; j < k0 + cnt; ++j) {
                orbit_total += c->load[j];
                if (c->load[j] > c->load[best]) best = j;
            }
            for (size_t j = k0; j < k0 + cnt; ++j) {
                if (c->quad_plain && chunks[j].scan == 1.0 && j == k0) {       // (plain order: chunk 0 of a four-cell orbit receives nothing)
                    in_lds[j] = 0;
                    continue;
                }
                if (j != best && c->load[j] < thr * orbit_total) {
                    in_lds[j] = 0;
                    duty[best] |= 1ull << (j - k0);
                }
            }
        }
    std::vector<double> cost(nc, 0.0);
    double total = 0;
    size_t n_lds = 0;
    (void)total;
    for (size_t k = 0; k < nc; ++k)
        if (in_lds[k]) {
            // an add costs ~12x less where the sums are 64-bit fixed point (ds_add_u64) than where they are fp32 (ds_add_f32)
            const double per_add = (c->n >= 4 && chunk_fixed(c, (int)chunks[k].variant)) ? add_cost * fixed_ratio : add_cost;
            cost[k] = chunks[k].scan * B + per_add * c->load[k];
            total += cost[k];
            ++n_lds;
        }
    // Feedback: where the last launches were timed (per-workgroup clocks, see replan), a chunk's cost is its measured
    // work; chunks without a measurement (they had no workgroup yet) keep the model's cost, scaled to the same unit.
    const bool use_work = c->work.size() == nc && c->n >= 4 && c->knob.plan_feedback != 0;
    if (use_work) {
        double measured = 0, modelled = 0;
        for (size_t k = 0; k < nc; ++k)
            if (in_lds[k] && c->work[k] > 0) {
                measured += c->work[k];
                modelled += cost[k];
            }
        if (measured > 0 && modelled > 0) {
            total = 0;
            for (size_t k = 0; k < nc; ++k)
                if (in_lds[k]) {
                    cost[k] = c->work[k] > 0 ? c->work[k] : cost[k] * measured / modelled;
                    total += cost[k];
                }
        }
    }
    std::vector<Slice> v;
    std::vector<uint32_t> parts(nc, 0);
    auto slice_of = [&](size_t k, uint32_t p, uint32_t np) {
        return Slice{chunks[k].variant, chunks[k].tlo, chunks[k].size, chunks[k].dlo, p, np, (uint32_t)k,
                     chunks[k].orb_tlo, chunks[k].orb_dlo, chunks[k].chunk0, duty[k], chunks[k].size >= OWN_SLOTS ? 15u : 14u,
                     chunk_fixed(c, (int)chunks[k].variant) ? 1u : 0u, 0u, c->quad_plain ? 1u : 0u};
    };
    // 0: flat plan; 1 (default where it applies): XCD-resident scan.  (Cutting the chunks into pieces packed onto the 32
    // workgroups of an XCD, a workgroup running its pieces one after the other, was tried: 0.296 -> 0.311 ms per step.)
    int xcd_plan = (c->n >= 4 && c->B >= (1u << 17) && n_lds <= CUS_PER_XCD) ? 1 : 0;
    if (!c->knob.plan_xcd) xcd_plan = 0;
    if (xcd_plan == 1) {
        // XCD-resident scan.  Workgroup i runs on XCD i % 8 and every XCD has its own 4 MB L2.  The records are cut into 8
        // ranges; range x is scanned only by the workgroups of XCD x (every chunk gets the same number of workgroups on each
        // XCD, so parts come in multiples of 8), and what fits of it stays in that L2 between the ~28 chunk scans instead of
        // every scan of every record going out to the fabric.  The coarser split balances a little worse than the flat
        // plan; with the measured costs it still wins: 0.3005 -> 0.2954 ms per step (n = 5), mean rule 0.400 -> 0.385.
        // Chunks too light for 8 workgroups (one per XCD is the finest the XCD-resident split allows) are scanned "flat"
        // instead: `fparts` workgroups anywhere, each taking every fparts-th record block.  Their scans miss the XCD's L2,
        // but they are the cheap ones (the centre square's records are 6 B), and the workgroups they give back go where the
        // makespan is: owner kernel 0.095 -> 0.085 ms at 2^20 lanes, n = 5 (G2048_PLAN_MIXED=0 turns it off).
        double lds_total = 0;
        for (size_t k = 0; k < nc; ++k)
            if (in_lds[k]) lds_total += cost[k];
        std::vector<uint32_t> fparts(nc, 0);
        uint32_t flat_wgs = 0, n_xcd = 0;
        const uint32_t budget = XCDS * CUS_PER_XCD;
        const uint32_t min_flat = (c->update_rule == 1 && c->knob.mean_one_pass) ? (uint32_t)((4ull * c->B) >> 21) + 1u : 1u;
        for (size_t k = 0; k < nc; ++k) {
            if (!in_lds[k]) continue;
            const double ideal = lds_total > 0 ? cost[k] / lds_total * budget : 8.0;
            if (ideal < c->knob.plan_mixed) {
                fparts[k] = std::max(min_flat, (uint32_t)(ideal + 0.999));
                flat_wgs += fparts[k];
            } else {
                ++n_xcd;
            }
        }
        if (n_xcd == 0 || flat_wgs + XCDS * n_xcd > budget) {        // (degenerate: fall back to all XCD-resident)
            std::fill(fparts.begin(), fparts.end(), 0u);
            flat_wgs = 0;
            n_xcd = (uint32_t)n_lds;
        }
        const uint32_t per_xcd_budget = (budget - flat_wgs) / XCDS;
        for (size_t k = 0; k < nc; ++k) parts[k] = (in_lds[k] && !fparts[k]) ? 1 : 0;
        for (size_t extra = n_xcd; extra < per_xcd_budget; ++extra) {       // greedy: the slowest workgroup gets help
            size_t worst = nc;
            for (size_t k = 0; k < nc; ++k)
                if (parts[k] && (worst == nc || cost[k] / parts[k] > cost[worst] / parts[worst])) worst = k;
            ++parts[worst];
        }
        for (uint32_t spare = budget - flat_wgs - XCDS * per_xcd_budget; spare > 0; --spare) {      // what does not make a group of 8
            size_t worst = nc;
            for (size_t k = 0; k < nc; ++k)
                if (fparts[k] && (worst == nc || cost[k] / fparts[k] > cost[worst] / fparts[worst])) worst = k;
            if (worst == nc) break;
            ++fparts[worst];
        }
        std::vector<std::pair<size_t, uint32_t>> per_xcd;                   // (chunk, j) of one XCD, longest-running first
        for (size_t k = 0; k < nc; ++k)
            for (uint32_t j = 0; j < parts[k]; ++j) per_xcd.push_back({k, j});
        std::stable_sort(per_xcd.begin(), per_xcd.end(), [&](const std::pair<size_t, uint32_t>& a, const std::pair<size_t, uint32_t>& b) {
            return cost[a.first] / parts[a.first] > cost[b.first] / parts[b.first];
        });
        for (const auto& kj : per_xcd)
            for (uint32_t x = 0; x < XCDS; ++x)
            {
                Slice sl = slice_of(kj.first, x * parts[kj.first] + kj.second, XCDS * parts[kj.first]);
                sl.xcd = 1u;
                v.push_back(sl);
            }
        for (size_t k = 0; k < nc; ++k)                                     // the flat ones behind the groups of 8
            for (uint32_t j = 0; j < fparts[k]; ++j) v.push_back(slice_of(k, j, fparts[k]));
        for (size_t k = 0; k < nc; ++k)
            if (fparts[k]) parts[k] = fparts[k];                            // (for the debug print)
    } else {
        // one-pass mean rule: a workgroup's count field must leave the sums enough bits (k_td_update_owner), so no
        // workgroup scans more than 2^21 / 4 records
        const uint32_t base_parts = (c->update_rule == 1 && c->knob.mean_one_pass) ? (uint32_t)((4ull * c->B) >> 21) + 1u : 1u;
        const uint32_t floor_wgs = (uint32_t)n_lds * base_parts;
        const uint32_t budget = c->B < (1u << 14) ? floor_wgs : (WG_BUDGET > floor_wgs ? WG_BUDGET : floor_wgs);
        for (size_t k = 0; k < nc; ++k) parts[k] = in_lds[k] ? base_parts : 0;
        for (size_t extra = floor_wgs; extra < budget; ++extra) {           // greedy: the slowest workgroup gets help
            size_t worst = nc;
            for (size_t k = 0; k < nc; ++k)
                if (in_lds[k] && (worst == nc || cost[k] / parts[k] > cost[worst] / parts[worst])) worst = k;
            ++parts[worst];
        }
        for (size_t k = 0; k < nc; ++k)
            for (uint32_t p = 0; p < parts[k]; ++p) v.push_back(slice_of(k, p, parts[k]));
        // longest-running workgroups first
        std::stable_sort(v.begin(), v.end(), [&](const Slice& a, const Slice& b) { return cost[a.chunk] / a.nparts > cost[b.chunk] / b.nparts; });
    }
    if (v.size() > MAX_SLICES) return fail(c, G2048_ERR_STATE, "LDS-owner plan too large");
    c->n_slices = (uint32_t)v.size();
    c->plan = v;
    if (c->knob.debug_plan) {
        fprintf(stderr, "[g2048 plan] %zu workgroups over %zu chunks; (chunk:load/parts)", v.size(), nc);
        for (size_t k = 0; k < nc; ++k) fprintf(stderr, " %zu:%.3f/%u", k, c->load[k] / (8.0 * B), parts[k]);
        fprintf(stderr, "\n");
    }
    // pinned staging, no wait: the copy is ordered before the next launch, and the staging is not rewritten before the
    // next replan's readback has synchronised the stream
    memcpy(c->h_slices, v.data(), v.size() * sizeof(Slice));
    HIP_TRY(c, hipMemcpyAsync(c->slices, c->h_slices, v.size() * sizeof(Slice), hipMemcpyHostToDevice, c->stream));
    c->steps_since_plan = 0;
    c->makespan_ref = 0;
    return G2048_OK;
}

// QuadOrder (see g2048_ctx): is chunk 0's share of the four-cell orbits' adds below the switch point?
bool quad_order_due(const g2048_ctx* c) {
    if (c->n < 4 || c->quad_plain || c->knob.quad_order != 0) return false;
    const std::vector<ChunkInfo> chunks = table_chunks(c);
    double hot = 0, all = 0;
    for (size_t k = 0; k < chunks.size() && k < c->load.size(); ++k) {
        if (chunks[k].scan != 1.0) continue;
        all += c->load[k];
        if (k == chunks[k].chunk0) hot += c->load[k];
    }
    return all > 0 && hot < c->knob.quad_switch * all;
}
// Switch the order in which the four-cell orbits' tables are used (between two steps: D is empty).  The loads measured under
// the old order mean nothing under the new one: back to a prior with the same totals, measured again from the next step on.
int set_quad_plain(g2048_ctx* c, bool plain) {
    const std::vector<ChunkInfo> chunks = table_chunks(c);
    for (size_t k0 = 0; k0 < chunks.size() && k0 < c->load.size(); k0 += chunks[k0].orb_chunks) {
        if (chunks[k0].scan != 1.0) continue;
        const size_t cnt = chunks[k0].orb_chunks;
        double total = 0;
        for (size_t j = k0; j < k0 + cnt; ++j) total += c->load[j];
        for (size_t j = k0; j < k0 + cnt; ++j) {
            const size_t rel = j - k0;
            c->load[j] = total * (plain ? (rel == 0 ? 0.0 : rel == 1 ? 0.7 : rel == 2 ? 0.28 : 0.01) : (rel == 0 ? 0.98 : 0.005));
        }
    }
    c->quad_plain = plain;
    c->work.clear();
    c->plan_measured = false;
    c->replan_interval = 1;
    if (c->knob.debug_plan) fprintf(stderr, "[g2048 plan] four-cell orbit tables: %s order from here on\n", plain ? "plain" : "hot-first");
    return build_slices(c);
}

// Planner feedback, first half (BEFORE the step's k_td_play is launched): the hit counters and the workgroup clocks of
// the update launches so far were stored into pinned host memory by the previous step's apply kernel (mirror_stats); an
// event marks that kernel's end.
int stats_readback(g2048_ctx* c) {
    if (c->n < 4 || c->n_chunks == 0 || c->steps_since_read == 0) return G2048_OK;
    if (c->knob.plan_poll && c->mirror_seq)
        c->mirror_want = c->mirror_seq;                         // the previous step's apply kernel announces them itself (mirror_stats)
    else
        HIP_TRY(c, hipEventRecord(c->ev_plan, c->stream));      // behind the previous step's apply kernel, which stored the statistics
    c->plan_pending = true;
    return G2048_OK;
}

// wait until the apply kernel with sequence number `want` has stored its statistics (it is queued in front of the k_td_play
// just launched: a fraction of a step away).  Bounded: a lost store must not hang the host.
bool mirror_arrived(const g2048_ctx* c, uint32_t want) {
    const volatile unsigned long long* tag = reinterpret_cast<const volatile unsigned long long*>(c->h_stat + STAT_BYTES);
    const uint32_t* h = reinterpret_cast<const uint32_t*>(c->h_stat);
    for (uint64_t spin = 0; spin < (1ull << 26); ++spin) {
        const unsigned long long t = *tag;          // {checksum : 32 | sequence number : 32}, one 8-byte store of the device
        if ((int32_t)((uint32_t)t - want) >= 0) {
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
            // trust the statistics only if they add up to the checksum that came with the sequence number: a set whose words
            // were still on their way when the tag landed does not (then keep polling — it is a fraction of a microsecond away)
            uint32_t sum = 0;
            for (size_t i = 0; i < STAT_BYTES / 4; ++i) sum += h[i];
            if (sum == (uint32_t)(t >> 32) || (uint32_t)t != want) return true;      // (a newer set than the one waited for: the caller reads what is there)
        }
        if ((spin & 1023u) == 1023u) std::this_thread::yield();
    }
    return false;
}

// Second half (AFTER k_td_play is launched, before the update's launch): wait for that copy — the GPU is busy with
// k_td_play for ~0.2 ms meanwhile — fold the counters into the load and the clocks into the per-chunk work, and rebuild
// the plan when it is due (1, 2, 4, ... replan_every steps after a start: the creation-time prior is wrong for whatever
// the boards look like, and one measured launch is enough to correct it) or when the last launch's makespan has grown
// past `imbalance` x what the plan achieved on its first launch (young synchronised boards change their tile
// distribution from one step to the next).
int replan(g2048_ctx* c) {
    if (!c->plan_pending) return G2048_OK;
    c->plan_pending = false;
    if (c->mirror_want) {
        const uint32_t want = c->mirror_want;
        c->mirror_want = 0;
        if (!mirror_arrived(c, want)) {         // (never seen; if it happens the context goes back to the event for good)
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            c->knob.plan_poll = 0;
        }
    } else {
        HIP_TRY(c, hipEventSynchronize(c->ev_plan));
    }
    const uint32_t* h = reinterpret_cast<const uint32_t*>(c->h_stat);
    if (c->discard_stats) {
        c->discard_stats = false;
        for (size_t k = 0; k < c->n_chunks; ++k) c->hits_seen[k] = h[k];
        c->steps_since_read = 0;
        return G2048_OK;
    }
    uint64_t fresh_total = 0;
    for (size_t k = 0; k < c->n_chunks; ++k) fresh_total += h[k] - c->hits_seen[k];          // cumulative counters, modulo 2^32
    // steps without records (the first move of fresh games makes none) say nothing about the load: keep what we have
    const bool informative = fresh_total >= (uint64_t)c->B / 4 * c->steps_since_read;
    for (size_t k = 0; k < c->n_chunks && informative; ++k) {
        const uint32_t fresh = h[k] - c->hits_seen[k];
        const double per_step = (double)fresh / c->steps_since_read;
        c->load[k] = c->plan_measured ? 0.5 * c->load[k] + 0.5 * per_step : per_step;
    }
    for (size_t k = 0; k < c->n_chunks; ++k) c->hits_seen[k] = h[k];
    c->steps_since_read = 0;
    if (!informative) {             // (the clocks of an empty launch say nothing either)
        c->steps_since_plan = 0;
        return G2048_OK;
    }
    c->plan_measured = true;
    if (quad_order_due(c)) return set_quad_plain(c, true);
    // the last launch's workgroup clocks: work of a chunk = (mean duration - fixed part) x its workgroups
    double makespan = 0;
    if (c->plan.size() == c->n_slices && c->n_slices) {
        const uint64_t* clk = reinterpret_cast<const uint64_t*>(c->h_stat + HITS_CAP * 4);
        const double fixed_ticks = 100.0 * c->knob.plan_fixed_us;          // (100 MHz clock)
        std::vector<double> sum(c->n_chunks, 0.0);
        std::vector<uint32_t> cnt(c->n_chunks, 0);
        uint64_t first = ~0ull, last = 0;
        for (uint32_t i = 0; i < c->n_slices; ++i) {
            const uint64_t a = clk[2 * i], b = clk[2 * i + 1];
            if (b <= a || b - a > 100000000ull || c->plan[i].chunk >= c->n_chunks) continue;      // (never launched / garbage)
            sum[c->plan[i].chunk] += (double)(b - a);
            ++cnt[c->plan[i].chunk];
            first = a < first ? a : first;
            last = b > last ? b : last;
        }
        if (last > first && last - first < 100000000ull) makespan = (double)(last - first);
        if (c->work.size() != c->n_chunks) c->work.assign(c->n_chunks, 0.0);
        for (uint32_t k = 0; k < c->n_chunks; ++k) {
            if (!cnt[k]) {
                c->work[k] = 0.0;                           // no workgroup: back to the model until it has one
                continue;
            }
            const double mean = sum[k] / cnt[k];
            const double w = (mean > fixed_ticks + 50.0 ? mean - fixed_ticks : 50.0) * cnt[k];
            c->work[k] = c->work[k] > 0 ? 0.5 * c->work[k] + 0.5 * w : w;
        }
    }
    const bool due = c->steps_since_plan >= c->replan_interval;
    const bool skew = c->makespan_ref > 0 && makespan > c->knob.imbalance * c->makespan_ref;
    if (c->makespan_ref == 0) c->makespan_ref = makespan;       // first measured launch under this plan
    if (c->knob.debug_plan)
        fprintf(stderr, "[g2048 feedback] steps under plan %u makespan %.1f us (ref %.1f)%s%s\n", c->steps_since_plan, makespan * 0.01,
                c->makespan_ref * 0.01, due ? " due" : "", skew ? " skew" : "");
    if (!due && !skew) return G2048_OK;
    if (due) c->replan_interval = std::min(c->replan_interval * 2u, std::max(1u, c->knob.replan_every));
    return build_slices(c);
}

// workgroups of k_td_play: as many as are resident at once (occupancy x CUs), or fewer if the batch is small
unsigned play_grid(g2048_ctx* c, bool hot) {
    const unsigned tpb = hot ? PLAY_HOT_WG : PLAY_WG;
    if (!c->play_wgs) {
        int per_cu = play_blocks_per_cu(c->n, hot), cus = 0;
        if (per_cu <= 0) per_cu = hot ? 1 : 2;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess || cus <= 0) cus = 256;
        c->play_wgs = (unsigned)(per_cu * cus);
        if (c->knob.play_wgs) c->play_wgs = c->knob.play_wgs;                                 // (experiments)
    }
    const unsigned need = (unsigned)(((uint64_t)c->B + tpb - 1) / tpb);
    return need < c->play_wgs ? need : c->play_wgs;
}

// how many lane blocks every workgroup of k_td_play takes without asking the counter: all full rounds but `play_dynamic`
uint32_t play_static_rounds(const g2048_ctx* c, unsigned grid, unsigned tpb) {
    const uint32_t nblocks = (uint32_t)(((uint64_t)c->B + 63) / 64), nwaves = grid * (tpb / 64), full = nblocks / nwaves;
    return full > c->knob.play_dynamic ? full - c->knob.play_dynamic : 0u;
}

// the LDS hot set pays once a workgroup has enough lanes to amortise its 88 KB copy; n >= 4 only
// (n = 4, 5 only.  Round 2 also built the n = 6 form behind G2048_PLAY_HOT=2: 68 of its 132 gathers are four-cell ones and go out
// in two batches of two directions; +3 % for a fresh agent, -6 % for a trained one, and it compiled to 256 VGPRs + 12 bytes of
// scratch — a spill between a hot gather's inline-asm load and its fence is exactly what that path must never have
// (tools/check_codeobj.py enforces it at build time), so the variant is gone.)
// n = 2, 3 (round 3): the whole table / its small-tile part in LDS (choose_small); their copies are smaller, so is the break-even
bool play_hot(const g2048_ctx* c) {
    if (c->knob.play_hot < 1) return false;
    if (c->n == 2) return c->B >= c->knob.play_hot_min / 16u;
    if (c->n == 3) return c->B >= c->knob.play_hot_min / 2u;
    return (c->n == 4 || c->n == 5) && c->B >= c->knob.play_hot_min;
}

// One TD step on the context's stream: k_td_play, then the update in the selected mode.  `ev` (optional) gets an
// event between the two parts.
int launch_td_step(g2048_ctx* c, float alpha, hipEvent_t ev = nullptr, hipEvent_t ev_owner = nullptr, hipEvent_t ev_tail = nullptr) {
    const uint32_t B = c->B;
    uint4* pn = c->prev[c->cur ^ 1];
    TdRecs recs;
    recs.state1 = c->prev[c->cur];
    recs.dw1 = c->dw1;
    recs.qstate = c->qstate;
    recs.qdw = c->qdw;
    recs.qcount = c->qcount + c->step_parity;
    recs.qcount_next = c->qcount + (c->step_parity ^ 1u);
    recs.dwmax = c->qcount + 2 + c->step_parity;
    recs.dwmax_next = c->qcount + 2 + (c->step_parity ^ 1u);
    recs.blocks = c->qcount + 32 + PLAY_SEGS * PLAY_SEG_STRIDE * c->step_parity;
    recs.blocks_next = c->qcount + 32 + PLAY_SEGS * PLAY_SEG_STRIDE * (c->step_parity ^ 1u);
    recs.unit = 0;
    recs.oidx = c->oidx[c->cur];
    recs.oidx_nxt = c->oidx[c->cur ^ 1];
    if (c->update_mode == 1) {
        const bool each = c->knob.feedback_each_step && B >= (1u << 17);
        if (each || c->steps_since_plan >= c->replan_interval)
            if (int rc = stats_readback(c)) return rc;
    }
    // lane re-ordering (LaneSort): when due, this step's k_td_play reads the lanes through the sorted permutation
    // (n = 3 since late round 3: nothing for a fresh agent, -9 % per step for a trained one; n = 2 has no table read outside LDS)
    const uint32_t* perm = nullptr;
    LaneSet lin = current_set(c), lout = lin;
    bool start_sort = false;            // take the keys behind this step's k_td_play and sort them on the side stream
    if (c->sort_every && c->n >= 3 && B >= c->knob.sort_min_batch) {
        if (c->sort_pending) {
            if (c->sort_wait == 0 || --c->sort_wait == 0) {
                HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_sort_done, 0));
                c->sort_pending = false;
                perm = c->sort_perm;
                lout = c->alt;
            }
        } else if (++c->steps_since_sort >= c->sort_every) {
            if (c->knob.sort_lag == 0) {
                if (int rc = lane_sort_permutation(c, false)) return rc;
                perm = c->sort_perm;
                lout = c->alt;
            } else {
                start_sort = true;
            }
        }
    }
    {
        const bool hot = play_hot(c);
        const unsigned grid = play_grid(c, hot);
        const hipError_t e = launch_td_play(c->stream, c->n, hot, grid, lin, lout, perm, pn, B, c->w, alpha, recs, c->auto_reset, c->stats, c->log,
                                            play_static_rounds(c, grid, hot ? PLAY_HOT_WG : PLAY_WG));
        if (e != hipSuccess) return fail(c, G2048_ERR_HIP, "k_td_play", e);
    }
    if (perm) {
        adopt_set(c, lout);
        c->alt = lin;
        c->permuted = true;
        c->steps_since_sort = 0;
    }
    if (start_sort) {
        if (int rc = lane_sort_permutation(c, true)) return rc;
        c->sort_pending = true;
        c->sort_wait = c->knob.sort_lag;
        c->steps_since_sort = 0;
    }
    if (ev) (void)hipEventRecord(ev, c->stream);
    if (c->update_mode == 1) {
        if (int rc = replan(c)) return rc;
        ++c->steps_since_plan;
        ++c->steps_since_read;
        const uint32_t tail_grid = B >= (1u << 16) ? 512 : 16;
        recs.unit = 0;
        // the LDS-owned orbit tables are double-buffered (k_apply_orbits clears the other one for the next step)
        const bool alt = c->dpar;
        float* Dcur = alt ? c->D2 : c->D;
        float* Doth = alt ? c->D : c->D2;
        float* Ccur = alt ? c->Dcnt2 : c->Dcnt;
        float* Coth = alt ? c->Dcnt : c->Dcnt2;
        float* dacc = c->tracking && c->knob.delta_accum ? c->delta : nullptr;
        // Mean rule: counts and sums come from ONE accumulation (count bits packed under the fixed-point sums) while the
        // count field leaves the sums 2^-18 of the largest |dw| as resolution: fewer than 2^21 adds per slot and workgroup
        // (the planner keeps the record ranges that short).
        // Beyond that (and for the n = 6 tail kernel) the counts take a second run of the same accumulation with dw = 1.
        bool one_pass = false;
        const bool hex_binned = c->n == 6 && c->knob.hex_bins && c->hex.pairs;
        if (c->update_rule == 1) {
            uint32_t min_parts = ~0u;
            for (const Slice& sl : c->plan) min_parts = sl.nparts < min_parts ? sl.nparts : min_parts;
            const uint64_t adds = (uint64_t)(c->n >= 4 ? 4 : 8) * ((B + min_parts - 1) / (min_parts ? min_parts : 1) + 8192);
            one_pass = c->knob.mean_one_pass && !c->plan.empty() && adds < (1ull << 21);
            TdRecs ones = recs;
            ones.unit = 1;
            if (!one_pass) {
                if (c->quad_plain || c->n < 4)
                    BY_N(c, (k_td_update_owner<N, true><<<c->n_slices, OWN_WG, 0, c->stream>>>(Ccur, nullptr, ones, B, c->slices, c->hits, c->wg_clock)))
                else
                    BY_N(c, (k_td_update_owner<N, false><<<c->n_slices, OWN_WG, 0, c->stream>>>(Ccur, nullptr, ones, B, c->slices, c->hits, c->wg_clock)))
            }
            if (c->n == 6 && !hex_binned) k_td_update_tail<6><<<tail_grid, OWN_WG, 0, c->stream>>>(c->Dcnt, ones, B, c->orbits.o[6].base, c->orbits.o[7].base);
        }
        float* dst = Dcur;
        float* cdst = one_pass ? Ccur : nullptr;
        if (c->quad_plain || c->n < 4)      // (n = 2, 3 have one form: the PLAIN instance)
            BY_N(c, (k_td_update_owner<N, true><<<c->n_slices, OWN_WG, 0, c->stream>>>(dst, cdst, recs, B, c->slices, c->hits, c->wg_clock)))
        else
            BY_N(c, (k_td_update_owner<N, false><<<c->n_slices, OWN_WG, 0, c->stream>>>(dst, cdst, recs, B, c->slices, c->hits, c->wg_clock)))
        if (ev_owner) (void)hipEventRecord(ev_owner, c->stream);
        if (hex_binned) {           // the f_6 orbits: bin the (slot, dw) pairs by table chunk, then LDS owners (k_hex_*)
            const uint32_t hb = c->orbits.o[6].base;
            const unsigned owners = (unsigned)std::min<uint64_t>(HEX_MAX_WORK, HEX_BINS + (uint64_t)2 * HEX_PAIRS * B / HEX_PART_PAIRS + 1);
            if (c->hex_young) {         // the opening of 2^20 synchronised games: exact runs (k_hex_count), see there
                --c->hex_young;
                HIP_TRY(c, hipMemsetAsync(c->hex.touched, 0, HEX_BINS * sizeof(uint32_t), c->stream));
                k_hex_count<6><<<512, OWN_WG, 0, c->stream>>>(recs, B, c->hex.touched);
                k_hex_layout_exact<<<1, OWN_WG, 0, c->stream>>>(c->hex, c->hex.touched);
            }
            k_hex_scatter<6><<<1024, OWN_WG, 0, c->stream>>>(recs, B, c->hex, c->D + hb, c->update_rule == 1 ? c->Dcnt + hb : nullptr);
            k_hex_plan<<<1, OWN_WG, 0, c->stream>>>(c->hex);
            k_hex_owner<<<owners, OWN_WG, 0, c->stream>>>(c->D + hb, c->update_rule == 1 ? c->Dcnt + hb : nullptr, recs, c->hex);
        } else if (c->n == 6) {
            k_td_update_tail<6><<<tail_grid, OWN_WG, 0, c->stream>>>(c->D, recs, B, c->orbits.o[6].base, c->orbits.o[7].base);
        }
        if (ev_tail) (void)hipEventRecord(ev_tail, c->stream);
        const StatMirror sm{reinterpret_cast<const uint32_t*>(c->statbuf), c->n >= 4 ? reinterpret_cast<uint32_t*>(c->h_stat_dev) : nullptr,
                            (uint32_t)(STAT_BYTES / 4), c->n >= 4 ? ++c->mirror_seq : 0u,
                            reinterpret_cast<unsigned long long*>(c->statbuf + STAT_BYTES)};
        OrbitTable ot = c->orbits;
        if (hex_binned) ot.total = c->owned_total;      // the f_6 orbit tables are applied chunk by chunk (k_hex_apply)
        if (c->update_rule == 1)
            k_apply_orbits_mean<<<grid_for(ot.total) + mirror_blocks(sm), WG, 0, c->stream>>>(c->w, dacc, c->D, c->Dcnt, Dcur, Ccur, Doth, Coth, c->owned_total, ot, sm, c->quad_plain ? 1u : 0u);
        else
            k_apply_orbits<<<grid_for(ot.total) + mirror_blocks(sm), WG, 0, c->stream>>>(c->w, dacc, c->D, Dcur, Doth, c->owned_total, ot, sm, c->quad_plain ? 1u : 0u);
        if (hex_binned) {
            const uint32_t hb = c->orbits.o[6].base;
            k_hex_apply<<<grid_for(HEX_SLOTS), WG, 0, c->stream>>>(c->w, dacc, c->D + hb, c->update_rule == 1 ? c->Dcnt + hb : nullptr, c->hex, c->orbits.o[6], c->orbits.o[7]);
        }
        c->dpar ^= 1u;
    } else {
        BY_N(c, (k_td_update<N><<<grid_for((uint64_t)B * 16), WG, 0, c->stream>>>(c->w, c->tracking && c->knob.delta_accum ? c->delta : nullptr, recs, B)));
        if (ev_owner) (void)hipEventRecord(ev_owner, c->stream);
        if (ev_tail) (void)hipEventRecord(ev_tail, c->stream);
    }
    c->cur ^= 1;
    c->step_parity ^= 1u;
    return G2048_OK;
}

int shape_of(int n, int* F, uint64_t* slots) {
    switch (n) {
        case 0: *F = 0; *slots = 0; return 0;
        case 2: *F = Shape<2>::F; *slots = Shape<2>::SLOTS; return 0;
        case 3: *F = Shape<3>::F; *slots = Shape<3>::SLOTS; return 0;
        case 4: *F = Shape<4>::F; *slots = Shape<4>::SLOTS; return 0;
        case 5: *F = Shape<5>::F; *slots = Shape<5>::SLOTS; return 0;
        case 6: *F = Shape<6>::F; *slots = Shape<6>::SLOTS; return 0;
        default: return -1;
    }
}

}  // namespace

extern "C" {

int g2048_abi_version(void) { return G2048_ABI_VERSION; }

const char* g2048_strerror(int s) {
    switch (s) {
        case G2048_OK: return "ok";
        case G2048_ERR_ARG: return "bad argument";
        case G2048_ERR_HIP: return "HIP runtime error";
        case G2048_ERR_NOMEM: return "out of memory";
        case G2048_ERR_STATE: return "invalid state for this call";
        case G2048_ERR_NODEV: return "no usable GPU";
        case G2048_ERR_COMM: return "RCCL error";
        default: return "unknown status";
    }
}

int g2048_device_count(int* count) {
    if (!count) return G2048_ERR_ARG;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    *count = e == hipSuccess ? n : 0;
    return G2048_OK;
}

int g2048_num_feat(int n) {
    int F;
    uint64_t s;
    return shape_of(n, &F, &s) == 0 && n != 0 ? F : G2048_ERR_ARG;
}

int64_t g2048_table_slots(int n) {
    int F;
    uint64_t s;
    return shape_of(n, &F, &s) == 0 && n != 0 ? (int64_t)s : (int64_t)G2048_ERR_ARG;
}

int g2048_feature_layout(int n, int64_t* offsets, int64_t* sizes) {
    int F;
    uint64_t s;
    if (shape_of(n, &F, &s) != 0 || n == 0) return G2048_ERR_ARG;
    for (int i = 0; i < F; ++i) {
        if (offsets) offsets[i] = feature_offset(n, i);
        if (sizes) sizes[i] = feature_size(n, i);
    }
    return G2048_OK;
}

const char* g2048_last_error(const g2048_ctx* c) { return c ? c->err.c_str() : "null context"; }

int g2048_destroy(g2048_ctx* c) {
    if (!c) return G2048_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->sort_stream) (void)hipStreamSynchronize(c->sort_stream);
    if (c->comm) (void)g2048_comm_destroy(c);
    if (c->parent) {            // the stream is drained: nothing of this context is pending on the shared table
        if (c->parent->shared_users) --c->parent->shared_users;
        if (c->parent->last_user == c) c->parent->last_user = nullptr;
    }
    void* bufs[] = {c->log.moves, c->log.start, c->log.final, c->log.meta, c->boards, c->scores, c->rng, c->prev[0], c->prev[1], c->oidx[0], c->oidx[1], c->label, c->flags, c->dw1, c->qstate,
                    c->qdw,    c->qcount, c->last_move, c->w,      c->w0,  c->delta,   c->stats,   c->scratch, c->slices, c->statbuf, c->D, c->Dcnt, c->D2, c->Dcnt2, c->pack, c->lane_id,
                    c->alt.boards, c->alt.scores, c->alt.rng, c->alt.label, c->alt.flags, c->alt.lane_id, c->alt.last_move,
                    c->sort_key16, c->sort_off, c->sort_perm, c->sort_cnt, c->sort_start,
                    c->hex.count, c->hex.base, c->hex.cursor, c->hex.pairs, c->hex.work, c->hex.nwork, c->hex.touched};
    for (void* p : bufs)
        if (p && (p != (void*)c->w || c->owns_table)) (void)hipFree(p);
    if (c->la_ws) (void)hipFree(c->la_ws);
    if (c->h_stat) (void)hipHostFree(c->h_stat);
    if (c->h_slices) (void)hipHostFree(c->h_slices);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev_plan) (void)hipEventDestroy(c->ev_plan);
    if (c->ev_table) (void)hipEventDestroy(c->ev_table);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->ev_sort_go) (void)hipEventDestroy(c->ev_sort_go);
    if (c->ev_sort_done) (void)hipEventDestroy(c->ev_sort_done);
    if (c->sort_stream) (void)hipStreamDestroy(c->sort_stream);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return G2048_OK;
}

static int create_impl(int device, uint32_t batch, int n_tuple, uint64_t seed, uint64_t lane0, g2048_ctx* parent, g2048_ctx** out) {
    if (!out) return G2048_ERR_ARG;
    *out = nullptr;
    int F;
    uint64_t slots;
    if (batch == 0 || batch > (1u << 28) || shape_of(n_tuple, &F, &slots) != 0) return G2048_ERR_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return G2048_ERR_NODEV;
    if (device < 0 || device >= ndev) return G2048_ERR_ARG;
    g2048_ctx* c = new (std::nothrow) g2048_ctx;
    if (!c) return G2048_ERR_NOMEM;
    c->device = device;
    c->B = batch;
    c->n = n_tuple;
    c->F = F;
    c->slots = slots;
    c->placed = n_tuple >= 4 ? slots : 0;
    c->seed = seed;
    c->lane0 = lane0;
    int rc = G2048_OK;
    auto bail = [&](int code) {
        // keep the message readable after the context is gone
        static thread_local std::string last;
        last = c->err;
        fprintf(stderr, "g2048_create: %s\n", last.c_str());
        g2048_destroy(c);
        return code;
    };
    if (hipSetDevice(device) != hipSuccess) return bail(G2048_ERR_HIP);
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return bail(G2048_ERR_HIP);
    if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_plan, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_table, hipEventDisableTiming) != hipSuccess)
        return bail(G2048_ERR_HIP);
    read_knobs(c);
    c->quad_plain = c->knob.quad_order == 2;
    // (n = 3: its LDS set holds the entries whose three cells are all below 8, so "big" for the lane order starts at 256 there)
    if (n_tuple == 3 && !getenv("G2048_SORT_TILE")) c->knob.sort_tile = 7;
    const size_t B = batch;
    if ((rc = dalloc(c, &c->boards, B)) || (rc = dalloc(c, &c->scores, B)) || (rc = dalloc(c, &c->rng, B)) ||
        (rc = dalloc(c, &c->prev[0], B)) || (rc = dalloc(c, &c->prev[1], B)) || (rc = dalloc(c, &c->label, B)) ||
        (rc = dalloc(c, &c->flags, B)) || (rc = dalloc(c, &c->dw1, B)) || (rc = dalloc(c, &c->qstate, B)) || (rc = dalloc(c, &c->qdw, B)) ||
        (rc = dalloc(c, &c->qcount, QCOUNT_WORDS)) || (rc = dalloc(c, &c->last_move, B)) || (rc = dalloc(c, &c->lane_id, B)) ||
        (rc = dalloc(c, &c->stats, 1)))
        return bail(rc);
    if (n_tuple >= 4 && ((rc = dalloc(c, &c->oidx[0], OIDX_BYTES_PER_LANE * B)) || (rc = dalloc(c, &c->oidx[1], OIDX_BYTES_PER_LANE * B)))) return bail(rc);
    if (parent) {
        g2048_ctx* root = parent->parent ? parent->parent : parent;
        c->w = root->w;
        c->owns_table = false;
        c->parent = root;
        // whatever the owner has queued on the table so far (weights_init, td_steps ...) comes first
        if (hipEventRecord(root->ev_table, root->stream) != hipSuccess) return bail(G2048_ERR_HIP);
        if (!root->last_user) root->last_user = root;
        ++root->shared_users;
    } else if (slots && (rc = dalloc(c, &c->w, slots))) {
        return bail(rc);
    }
    if ((rc = build_slices(c))) return bail(rc);
    if (hipMemsetAsync(c->stats, 0, sizeof(Stats), c->stream) != hipSuccess ||
        hipMemsetAsync(c->prev[0], 0, B * sizeof(uint4), c->stream) != hipSuccess ||
        hipMemsetAsync(c->prev[1], 0, B * sizeof(uint4), c->stream) != hipSuccess ||
        hipMemsetAsync(c->qcount, 0, QCOUNT_WORDS * 4, c->stream) != hipSuccess || hipMemsetAsync(c->last_move, 0, B * 2, c->stream) != hipSuccess || hipMemsetAsync(c->dw1, 0, B * 4, c->stream) != hipSuccess ||
        (slots && !parent && hipMemsetAsync(c->w, 0, slots * sizeof(float), c->stream) != hipSuccess))
        return bail(G2048_ERR_HIP);
    k_iota<<<grid_for(B), WG, 0, c->stream>>>(c->lane_id, batch);
    c->sort_every = c->knob.sort_every;
    k_seed<<<grid_for(B), WG, 0, c->stream>>>(c->rng, batch, seed, lane0);
    k_new_games<<<grid_for(B), WG, 0, c->stream>>>(c->boards, c->scores, c->rng, c->label, c->flags, batch);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) {
        c->err = "initial kernels failed";
        return bail(G2048_ERR_HIP);
    }
    *out = c;
    return G2048_OK;
}

int g2048_create(int device, uint32_t batch, int n_tuple, uint64_t seed, uint64_t lane0, g2048_ctx** out) {
    return create_impl(device, batch, n_tuple, seed, lane0, nullptr, out);
}

int g2048_create_shared(g2048_ctx* parent, uint32_t batch, uint64_t seed, uint64_t lane0, g2048_ctx** out) {
    if (!parent || parent->n == 0) return G2048_ERR_ARG;
    return create_impl(parent->device, batch, parent->n, seed, lane0, parent, out);
}

int g2048_sync(g2048_ctx* c) {
    if (!c) return G2048_ERR_ARG;
    if (int rc = bind(c)) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return G2048_OK;
}

int g2048_timer_start(g2048_ctx* c) {
    if (!c) return G2048_ERR_ARG;
    if (int rc = bind(c)) return rc;
    HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
    return G2048_OK;
}

int g2048_timer_stop(g2048_ctx* c, float* ms) {
    if (!c || !ms) return G2048_ERR_ARG;
    if (int rc = bind(c)) return rc;
    HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
    HIP_TRY(c, hipEventSynchronize(c->ev1));
    HIP_TRY(c, hipEventElapsedTime(ms, c->ev0, c->ev1));
    return G2048_OK;
}

#define IO_PAIR(name, field, type, per)                                                            \
    int g2048_set_##name(g2048_ctx* c, const type* src) {                                          \
        if (!c || !src) return c ? fail(c, G2048_ERR_ARG, "null buffer") : G2048_ERR_ARG;          \
        if (int rc = bind(c)) return rc;                                                           \
        NEED_IDENTITY(c);                                                                          \
        return h2d(c, c->field, src, (size_t)c->B * (per) * sizeof(type));                         \
    }                                                                                              \
    int g2048_get_##name(g2048_ctx* c, type* dst) {                                                \
        if (!c || !dst) return c ? fail(c, G2048_ERR_ARG, "null buffer") : G2048_ERR_ARG;          \
        if (int rc = bind(c)) return rc;                                                           \
        NEED_IDENTITY(c);                                                                          \
        return d2h(c, dst, c->field, (size_t)c->B * (per) * sizeof(type));                         \
    }

IO_PAIR(boards, boards, uint8_t, 16)
IO_PAIR(scores, scores, int32_t, 1)
IO_PAIR(rng, rng, uint64_t, 2)

int g2048_get_carry(g2048_ctx* c, uint8_t* prev, float* label, uint8_t* flags) {
    if (!c) return G2048_ERR_ARG;
    if (int rc = bind(c)) return rc;
    NEED_IDENTITY(c);
    int rc = G2048_OK;
    if (prev) {
        if ((rc = d2h(c, prev, c->prev[c->cur], (size_t)c->B * 16))) return rc;
        for (size_t i = 0; i < c->B; ++i) {           // packed rows -> 16 tile bytes (host-side format conversion only)
            uint32_t v[4];
            memcpy(v, prev + 16 * i, 16);
            const uint32_t rows[4] = {v[0] & 0xFFFFu, v[0] >> 16, v[1] & 0xFFFFu, v[1] >> 16};
            for (int r = 0; r < 4; ++r)
                for (int col = 0; col < 4; ++col) prev[16 * i + 4 * r + col] = (uint8_t)((rows[r] >> (12 - 4 * col)) & 0xFu);
        }
    }
    if (label && (rc = d2h(c, label, c->label, (size_t)c->B * 4))) return rc;
    if (flags && (rc = d2h(c, flags, c->flags, (size_t)c->B))) return rc;
    return rc;
}

int g2048_clear_carry(g2048_ctx* c) {
    if (!c) return G2048_ERR_ARG;
    if (int rc = bind(c)) return rc;
    HIP_TRY(c, hipMemsetAsync(c->flags, 0, c->B, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->label, 0, (size_t)c->B * 4, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->dw1, 0, (size_t)c->B * 4, c->stream));
    return G2048_OK;
}

int g2048_reset(g2048_ctx* c) {
    if (!c) return G2048_ERR_ARG;
    if (int rc = bind(c)) return rc;
    NEED_IDENTITY(c);
    c->replan_interval = 1;         // the tile distribution restarts: follow it closely again
    c->hex_young = HEX_YOUNG_STEPS;
    if (c->n >= 4 && c->quad_plain && c->knob.quad_order == 0 && c->slices) {
        c->discard_stats = true;
        if (int rc = set_quad_plain(c, false)) return rc;       // young boards again: hot-first (QuadOrder)
    }
    k_new_games<<<grid_for(c->B), WG, 0, c->stream>>>(c->boards, c->scores, c->rng, c->label, c->flags, c->B);
    if (c->log.lanes) k_log_init<<<grid_for(c->log.lanes), WG, 0, c->stream>>>(c->log, c->boards, c->flags);
    return launched(c, "k_new_games");
}

int g2048_set_auto_reset(g2048_ctx* c, int on) {
    if (!c) return G2048_ERR_ARG;
    c->auto_reset = on ? 1 : 0;
    return G2048_OK;
}

int g2048_move_all(g2048_ctx* c, uint8_t* after, int32_t* reward, uint8_t* changed) {
    if (!c || !after || !reward || !changed) return c ? fail(c, G2048_ERR_ARG, "null buffer") : G2048_ERR_ARG;
    if (int rc = bind(c)) return rc;
    NEED_IDENTITY(c);
    const size_t B = c->B;
    if (int rc = ensure_scratch(c, B * (64 + 16 + 1))) return rc;
    uint4* d_after = (uint4*)c->scratch;
    int4* d_reward = (int4*)((char*)c->scratch + B * 64);
    uint8_t* d_changed = (uint8_t*)c->scratch + B * 80;
    k_move_all<<<grid_for(B), WG, 0, c->stream>>>(c->boards, c->B, d_after, d_reward, d_changed);
    if (int rc = launched(c, "k_move_all")) return rc;
    int rc;
    if ((rc = d2h(c, after, d_after, B * 64)) || (rc = d2h(c, reward, d_reward, B * 16)) || (rc = d2h(c, changed, d_changed, B))) return rc;
    return G2048_OK;
}

int g2048_apply_moves(g2048_ctx* c, const uint8_t* dirs, uint8_t* moved) {
    if (!c || !dirs) return c ? fail(c, G2048_ERR_ARG, "null buffer") : G2048_ERR_ARG;
    if (int rc = bind(c)) return rc;
    NEED_IDENTITY(c);
    const size_t B = c->B;
    if (int rc = ensure_scratch(c, 2 * B)) return rc;
    uint8_t* d_dirs = (uint8_t*)c->scratch;
    uint8_t* d_moved = d_dirs + B;
    if (int rc = h2d(c, d_dirs, dirs, B)) return rc;
    k_apply_moves<<<grid_for(B), WG, 0, c->stream>>>(c->boards, c->scores, c->B, d_dirs, d_moved);
    if (int rc = launched(c, "k_apply_moves")) return rc;
    if (moved) return d2h(c, moved, d_moved, B);
    return G2048_OK;
}

int g2048_terminal(g2048_ctx* c, uint8_t* over, uint8_t* n_empty, uint8_t* n_pairs) {
    if (!c) return G2048_ERR_ARG;
    if (int rc = bind(c)) return rc;
    NEED_IDENTITY(c);
    const size_t B = c->B;
    if (int rc = ensure_scratch(c, 3 * B)) return rc;
    uint8_t* d = (uint8_t*)c->scratch;
    k_terminal<<<grid_for(B), WG, 0, c->stream>>>(c->boards, c->B, d, d + B, d + 2 * B);
    if (int rc = launched(c, "k_terminal")) return rc;
    int rc;
    if (over && (rc = d2h(c, over, d, B))) return rc;
    if (n_empty && (rc = d2h(c, n_empty, d + B, B))) return rc;
    if (n_pairs && (rc = d2h(c, n_pairs, d + 2 * B, B))) return rc;
    return g2048_sync(c);
}

static int spawn_impl(g2048_ctx* c, const uint8_t* in_r10, const uint8_t* in_k, uint8_t* out_r10, uint8_t* out_k) {
    if (int rc = bind(c)) return rc;
    NEED_IDENTITY(c);
    const size_t B = c->B;
    if (int rc = ensure_scratch(c, 4 * B)) return rc;
    uint8_t* d = (uint8_t*)c->scratch;
    if (in_r10) {
        int rc;
        if ((rc = h2d(c, d, in_r10, B)) || (rc = h2d(c, d + B, in_k, B))) return rc;
    }
    k_spawn<<<grid_for(B), WG, 0, c->stream>>>(c->boards, c->rng, c->B, in_r10 ? d : nullptr, in_r10 ? d + B : nullptr, d + 2 * B, d + 3 * B);
    if (int rc = launched(c, "k_spawn")) return rc;
    int rc;
    if (out_r10 && (rc = d2h(c, out_r10, d + 2 * B, B))) return rc;
    if (out_k && (rc = d2h(c, out_k, d + 3 * B, B))) return rc;
    return G2048_OK;
}

int g2048_spawn(g2048_ctx* c, uint8_t* r10, uint8_t* k) {
    if (!c) return G2048_ERR_ARG;
    return spawn_impl(c, nullptr, nullptr, r10, k);
}

int g2048_spawn_injected(g2048_ctx* c, const uint8_t* r10, const uint8_t* k) {
    if (!c || !r10 || !k) return c ? fail(c, G2048_ERR_ARG, "null buffer") : G2048_ERR_ARG;
    return spawn_impl(c, r10, k, nullptr, nullptr);
}

// ---- stateless forms: any number of caller-supplied boards, the lanes are not touched (Game.pre_move / game_over and
// QAgent.evaluate on arbitrary positions, e.g. the nodes of the look-ahead tree, game_logic.py:214-243)
int g2048_boards_move_all(g2048_ctx* c, const uint8_t* boards, int64_t count, uint8_t* after, int32_t* reward, uint8_t* changed) {
    if (!c || !boards || !after || !reward || !changed) return c ? fail(c, G2048_ERR_ARG, "null buffer") : G2048_ERR_ARG;
    NEED(c, count >= 0 && count <= (1 << 28), "bad board count");
    if (count == 0) return G2048_OK;
    if (int rc = bind(c)) return rc;
    const size_t n = (size_t)count;
    if (int rc = ensure_scratch(c, n * (16 + 64 + 16 + 1))) return rc;
    uint4* d_boards = (uint4*)c->scratch;
    uint4* d_after = d_boards + n;
    int4* d_reward = (int4*)(d_after + 4 * n);
    uint8_t* d_changed = (uint8_t*)(d_reward + n);
    int rc;
    if ((rc = h2d(c, d_boards, boards, n * 16))) return rc;
    k_move_all<<<grid_for(n), WG, 0, c->stream>>>(d_boards, (uint32_t)n, d_after, d_reward, d_changed);
    if ((rc = launched(c, "k_move_all"))) return rc;
    if ((rc = d2h(c, after, d_after, n * 64)) || (rc = d2h(c, reward, d_reward, n * 16)) || (rc = d2h(c, changed, d_changed, n))) return rc;
    return G2048_OK;
}

int g2048_boards_evaluate(g2048_ctx* c, const uint8_t* boards, int64_t count, float* value) {
    if (!c || !boards || !value) return c ? fail(c, G2048_ERR_ARG, "null buffer") : G2048_ERR_ARG;
    NEED_TABLE(c);
    NEED(c, count >= 0 && count <= (1 << 28), "bad board count");
    if (count == 0) return G2048_OK;
    if (int rc = bind(c)) return rc;
    USE_TABLE(c);
    const size_t n = (size_t)count;
    if (int rc = ensure_scratch(c, n * 20)) return rc;
    uint4* d_boards = (uint4*)c->scratch;
    float* d_value = (float*)(d_boards + n);
    int rc;
    if ((rc = h2d(c, d_boards, boards, n * 16))) return rc;
    BY_N(c, (k_evaluate<N><<<grid_for(n), WG, 0, c->stream>>>(d_boards, (uint32_t)n, c->w, d_value)));
    if ((rc = launched(c, "k_evaluate"))) return rc;
    return d2h(c, value, d_value, n * 4);
}

// ---- look-ahead (lookahead.hip)

namespace {
int ensure_la_workspace(g2048_ctx* c, size_t bytes) {
    if (bytes <= c->la_ws_bytes) return G2048_OK;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->la_ws) HIP_TRY(c, hipFree(c->la_ws));
    c->la_ws = nullptr;
    c->la_ws_bytes = 0;
    hipError_t e = hipMalloc(&c->la_ws, bytes);
    if (e != hipSuccess) return fail(c, G2048_ERR_NOMEM, "hipMalloc(look-ahead workspace)", e);
    c->la_ws_bytes = bytes;
    return G2048_OK;
}

int la_plan(g2048_ctx* c, int depth, int width, int since_empty, int limit_tile, LaPlan* plan) {
    NEED(c, depth >= 0 && depth <= LA_MAX_DEPTH, "look-ahead depth out of range (0 .. 6)");
    NEED(c, width >= 1 && width <= LA_MAX_WIDTH, "look-ahead width out of range (1 .. 16)");
    NEED(c, since_empty >= 0 && limit_tile >= 0, "negative since_empty / limit_tile");
    NEED(c, la_leaves_per_root(depth, width) != 0, "look-ahead tree too large: (4 width)^depth must stay below 2^22 nodes per position");
    *plan = LaPlan{c->n, depth, width, since_empty > 255 ? 255 : since_empty, limit_tile > 255 ? 255 : limit_tile};
    return G2048_OK;
}
}  // namespace

int g2048_boards_look_forward(g2048_ctx* c, const uint8_t* boards, int64_t count, int depth, int width, int since_empty, const uint64_t* salt, float* value) {
    if (!c || !boards || !value) return c ? fail(c, G2048_ERR_ARG, "null buffer") : G2048_ERR_ARG;
    NEED_TABLE(c);
    NEED(c, count >= 0 && count <= (1 << 26), "bad board count");
    LaPlan plan;
    if (int rc = la_plan(c, depth, width, since_empty, 0, &plan)) return rc;
    if (count == 0) return G2048_OK;
    if (int rc = bind(c)) return rc;
    USE_TABLE(c);
    const size_t n = (size_t)count;
    const uint64_t round = std::min<uint64_t>(la_roots_per_round(depth, width), n);
    if (int rc = ensure_la_workspace(c, la_workspace_bytes(round, depth, width))) return rc;
    if (int rc = ensure_scratch(c, n * (16 + 16 + 4))) return rc;
    uint4* d_boards = (uint4*)c->scratch;
    ulonglong2* d_salt = (ulonglong2*)(d_boards + n);
    float* d_value = (float*)(d_salt + n);
    int rc;
    if ((rc = h2d(c, d_boards, boards, n * 16))) return rc;
    if (salt && (rc = h2d(c, d_salt, salt, n * 16))) return rc;
    const hipError_t e = la_values(c->stream, plan, c->w, d_boards, salt ? d_salt : nullptr, n, c->la_ws, c->la_ws_bytes, d_value);
    if (e != hipSuccess) return fail(c, G2048_ERR_HIP, "look-ahead kernels", e);
    return d2h(c, value, d_value, n * 4);
}

int g2048_lookahead_steps(g2048_ctx* c, int depth, int width, int since_empty, int limit_tile, uint32_t nsteps) {
    if (!c) return G2048_ERR_ARG;
    NEED_TABLE(c);
    LaPlan plan;
    if (int rc = la_plan(c, depth, width, since_empty, limit_tile, &plan)) return rc;
    if (int rc = bind(c)) return rc;
    NEED_IDENTITY(c);
    USE_TABLE(c);
    const uint64_t round = std::min<uint64_t>(la_roots_per_round(depth, width), (uint64_t)c->B * 4);
    if (int rc = ensure_la_workspace(c, la_workspace_bytes(round, depth, width))) return rc;
    const hipError_t e = la_steps(c->stream, plan, c->w, current_set(c), c->B, c->auto_reset, c->stats, c->log, c->la_ws, c->la_ws_bytes, nsteps);
    if (e != hipSuccess) return fail(c, G2048_ERR_HIP, "look-ahead kernels", e);
    return G2048_OK;
}

int g2048_step_random(g2048_ctx* c, uint32_t nsteps) {
    if (!c) return G2048_ERR_ARG;
    if (int rc = bind(c)) return rc;
    k_step_random<<<grid_for(c->B), WG, 0, c->stream>>>(c->boards, c->scores, c->rng, c->flags, c->B, nsteps, c->auto_reset, c->stats);
    return launched(c, "k_step_random");
}

int g2048_features(g2048_ctx* c, int32_t* out) {
    if (!c || !out) return c ? fail(c, G2048_ERR_ARG, "null buffer") : G2048_ERR_ARG;
    NEED_TABLE(c);
    if (int rc = bind(c)) return rc;
    NEED_IDENTITY(c);
    const size_t bytes = (size_t)c->B * c->F * 4;
    if (int rc = ensure_scratch(c, bytes)) return rc;
    BY_N(c, (k_features<N><<<grid_for(c->B), WG, 0, c->stream>>>(c->boards, c->B, (int32_t*)c->scratch)));
    if (int rc = launched(c, "k_features")) return rc;
    return d2h(c, out, c->scratch, bytes);
}

int g2048_weights_set(g2048_ctx* c, const float* w, int64_t count) {
    if (!c || !w) return c ? fail(c, G2048_ERR_ARG, "null buffer") : G2048_ERR_ARG;
    NEED_TABLE(c);
    NEED(c, count == (int64_t)c->slots, "weight count does not match the table");
    if (int rc = bind(c)) return rc;
    USE_TABLE(c);
    if (!c->placed) return h2d(c, c->w, w, c->slots * 4);
    float* flat = nullptr;                          // index order -> memory order through a staging buffer
    hipError_t e = hipMalloc(&flat, c->slots * 4);
    if (e != hipSuccess) return fail(c, G2048_ERR_NOMEM, "hipMalloc(table staging)", e);
    int rc = h2d(c, flat, w, c->slots * 4);
    if (!rc) {
        k_table_in<<<2048, WG, 0, c->stream>>>(c->w, flat, c->slots);
        rc = launched(c, "k_table_in");
    }
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(flat);
    return rc;
}

int g2048_weights_get(g2048_ctx* c, float* w, int64_t count) {
    if (!c || !w) return c ? fail(c, G2048_ERR_ARG, "null buffer") : G2048_ERR_ARG;
    NEED_TABLE(c);
    NEED(c, count == (int64_t)c->slots, "weight count does not match the table");
    if (int rc = bind(c)) return rc;
    USE_TABLE(c);
    if (!c->placed) return d2h(c, w, c->w, c->slots * 4);
    float* flat = nullptr;                          // memory order -> index order through a staging buffer
    hipError_t e = hipMalloc(&flat, c->slots * 4);
    if (e != hipSuccess) return fail(c, G2048_ERR_NOMEM, "hipMalloc(table staging)", e);
    k_delta_out<<<2048, WG, 0, c->stream>>>(c->w, flat, c->slots, c->placed);
    int rc = launched(c, "k_delta_out");
    if (!rc) rc = d2h(c, w, flat, c->slots * 4);
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(flat);
    return rc;
}

int g2048_weights_init(g2048_ctx* c, uint64_t seed, float scale) {
    if (!c) return G2048_ERR_ARG;
    NEED_TABLE(c);
    if (int rc = bind(c)) return rc;
    USE_TABLE(c);
    k_weights_init<<<2048, WG, 0, c->stream>>>(c->w, c->slots, c->placed, seed, scale);
    return launched(c, "k_weights_init");
}

int g2048_evaluate(g2048_ctx* c, float* value) {
    if (!c || !value) return c ? fail(c, G2048_ERR_ARG, "null buffer") : G2048_ERR_ARG;
    NEED_TABLE(c);
    if (int rc = bind(c)) return rc;
    NEED_IDENTITY(c);
    USE_TABLE(c);
    if (int rc = ensure_scratch(c, (size_t)c->B * 4)) return rc;
    BY_N(c, (k_evaluate<N><<<grid_for(c->B), WG, 0, c->stream>>>(c->boards, c->B, c->w, (float*)c->scratch)));
    if (int rc = launched(c, "k_evaluate")) return rc;
    return d2h(c, value, c->scratch, (size_t)c->B * 4);
}

int g2048_eval_select(g2048_ctx* c, float* value, uint8_t* action, float* values4) {
    if (!c || ((value == nullptr) != (action == nullptr))) return c ? fail(c, G2048_ERR_ARG, "null buffer") : G2048_ERR_ARG;
    NEED_TABLE(c);
    if (int rc = bind(c)) return rc;
    NEED_IDENTITY(c);
    USE_TABLE(c);
    const size_t B = c->B;
    if (int rc = ensure_scratch(c, B * (16 + 4 + 1))) return rc;
    float4* d_v4 = (float4*)c->scratch;
    float* d_v = (float*)((char*)c->scratch + B * 16);
    uint8_t* d_a = (uint8_t*)c->scratch + B * 20;
    // n = 2, 3 on batches big enough to pay for the copy: the LDS form, one workgroup per CU
    const unsigned lds_grid = (unsigned)std::min<uint64_t>(256, (B + PLAY_HOT_WG - 1) / PLAY_HOT_WG);
    const bool lds = c->knob.play_hot >= 1 && ((c->n == 2 && B >= (1u << 14)) || (c->n == 3 && B >= (1u << 16)));
    {
        const hipError_t e = launch_eval_select(c->stream, c->n, lds, lds_grid, c->boards, c->B, c->w, d_v, d_a, values4 ? d_v4 : nullptr);
        if (e != hipSuccess) return fail(c, G2048_ERR_HIP, "k_eval_select", e);
    }
    if (int rc = launched(c, "k_eval_select")) return rc;
    if (!value) return G2048_OK;            // device-only run (results stay in the context's scratch buffer; benchmarking)
    int rc;
    if ((rc = d2h(c, value, d_v, B * 4)) || (rc = d2h(c, action, d_a, B))) return rc;
    if (values4 && (rc = d2h(c, values4, d_v4, B * 16))) return rc;
    return G2048_OK;
}

int g2048_update(g2048_ctx* c, const uint8_t* states, const float* dw, int64_t count) {
    if (!c || !states || !dw) return c ? fail(c, G2048_ERR_ARG, "null buffer") : G2048_ERR_ARG;
    NEED_TABLE(c);
    NEED(c, count >= 0 && count <= (1 << 28), "bad record count");
    if (count == 0) return G2048_OK;
    if (int rc = bind(c)) return rc;
    USE_TABLE(c);
    const size_t n = (size_t)count;
    if (int rc = ensure_scratch(c, n * 20)) return rc;
    uint4* d_states = (uint4*)c->scratch;
    float* d_dw = (float*)((char*)c->scratch + n * 16);
    int rc;
    if ((rc = h2d(c, d_states, states, n * 16)) || (rc = h2d(c, d_dw, dw, n * 4))) return rc;
    BY_N(c, (k_update_records<N><<<grid_for(n * 8), WG, 0, c->stream>>>(c->w, c->tracking && c->knob.delta_accum ? c->delta : nullptr, d_states, d_dw, (uint32_t)n)));
    if ((rc = launched(c, "k_update_records"))) return rc;
    return g2048_sync(c);
}

int g2048_td_steps(g2048_ctx* c, float alpha, uint32_t nsteps) {
    if (!c) return G2048_ERR_ARG;
    NEED_TABLE(c);
    if (int rc = bind(c)) return rc;
    USE_TABLE(c);
    for (uint32_t s = 0; s < nsteps; ++s)
        if (int rc = launch_td_step(c, alpha)) return rc;
    return launched(c, "k_td_play/k_td_update");
}

int g2048_set_lane_sort(g2048_ctx* c, uint32_t every) {
    if (!c) return G2048_ERR_ARG;
    c->sort_every = every;
    c->steps_since_sort = 0;
    c->sort_pending = false;
    return G2048_OK;
}

// test hook: run the lane re-order on the current boards, in line, and hand back what it produced — perm (position i of the
// new order takes the lane at position perm[i]) and every position's key.  The lane order itself is left as it is.
int g2048_debug_lane_order(g2048_ctx* c, uint32_t* perm, uint16_t* keys) {
    if (!c || !perm || !keys) return c ? fail(c, G2048_ERR_ARG, "null output") : G2048_ERR_ARG;
    if (int rc = bind(c)) return rc;
    if (int rc = lane_sort_permutation(c, false)) return rc;
    if (int rc = d2h(c, perm, c->sort_perm, (size_t)c->B * 4)) return rc;
    return d2h(c, keys, c->sort_key16, (size_t)c->B * 2);
}

int g2048_set_update_mode(g2048_ctx* c, int mode) {
    if (!c) return G2048_ERR_ARG;
    NEED(c, mode == 0 || mode == 1, "update mode must be 0 (global atomics) or 1 (LDS-owner)");
    NEED(c, mode == 1 || c->update_rule == 0, "the per-slot mean rule needs update mode 1");
    c->update_mode = mode;
    return G2048_OK;
}

int g2048_set_update_rule(g2048_ctx* c, int rule) {
    if (!c) return G2048_ERR_ARG;
    NEED_TABLE(c);
    NEED(c, rule == 0 || rule == 1, "update rule must be 0 (add every dw) or 1 (per-slot mean)");
    if (rule == 1) {
        NEED(c, c->update_mode == 1, "the per-slot mean rule needs the LDS-owner update (update mode 1)");
        if (int rc = bind(c)) return rc;
        const size_t count = c->orbits.total;
        int rc;
        if (!c->D) {
            if ((rc = dalloc(c, &c->D, count))) return rc;
            HIP_TRY(c, hipMemset(c->D, 0, count * 4));
        }
        if (!c->Dcnt) {
            if ((rc = dalloc(c, &c->Dcnt, count))) return rc;
            HIP_TRY(c, hipMemset(c->Dcnt, 0, count * 4));
        }
        if (!c->Dcnt2) {
            if ((rc = dalloc(c, &c->Dcnt2, c->owned_total))) return rc;
            HIP_TRY(c, hipMemset(c->Dcnt2, 0, (size_t)c->owned_total * 4));
        }
        if (c->update_rule == 0) {     // steps under the sum rule leave the count buffers alone: start clean
            HIP_TRY(c, hipMemsetAsync(c->Dcnt, 0, (size_t)c->owned_total * 4, c->stream));
            HIP_TRY(c, hipMemsetAsync(c->Dcnt2, 0, (size_t)c->owned_total * 4, c->stream));
        }
    }
    const bool rechunk = c->n >= 4 && c->update_rule != rule;      // the cross orbit changes its chunk size with the rule
    c->update_rule = rule;
    if (rechunk) {
        if (int rc = bind(c)) return rc;
        HIP_TRY(c, hipMemsetAsync(c->statbuf, 0, STAT_BYTES, c->stream));
        c->n_chunks = 0;
        c->work.clear();
        c->plan_measured = false;
        c->replan_interval = 1;
        c->steps_since_read = 0;
        return build_slices(c);
    }
    return G2048_OK;
}

// average milliseconds per launch: [0] k_td_play, [1] k_td_update_owner (both passes under the mean rule; k_td_update in
// update mode 0), [2] k_td_update_tail (n = 6, else 0), [3] k_apply_* — HIP events on the context's stream, one wait per step
int g2048_td_steps_kernel_ms(g2048_ctx* c, float alpha, uint32_t nsteps, float* out4) {
    if (!c || !out4) return c ? fail(c, G2048_ERR_ARG, "null buffer") : G2048_ERR_ARG;
    NEED_TABLE(c);
    NEED(c, nsteps <= 256, "at most 256 steps per call");
    if (int rc = bind(c)) return rc;
    USE_TABLE(c);
    // five events per step, all read after ONE wait at the end: the steps run back to back as in g2048_td_steps
    std::vector<hipEvent_t> e(5 * (size_t)nsteps, nullptr);
    int rc = G2048_OK;
    for (auto& ev : e)
        if (rc == G2048_OK && hipEventCreate(&ev) != hipSuccess) rc = fail(c, G2048_ERR_HIP, "hipEventCreate");
    for (uint32_t s = 0; s < nsteps && rc == G2048_OK; ++s) {
        hipEvent_t* v = &e[5 * (size_t)s];
        (void)hipEventRecord(v[0], c->stream);
        rc = launch_td_step(c, alpha, v[1], v[2], v[3]);
        (void)hipEventRecord(v[4], c->stream);
    }
    if (rc == G2048_OK && hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(c, G2048_ERR_HIP, "hipStreamSynchronize");
    double t[4] = {0, 0, 0, 0};
    for (uint32_t s = 0; s < nsteps && rc == G2048_OK; ++s)
        for (int j = 0; j < 4 && rc == G2048_OK; ++j) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, e[5 * (size_t)s + j], e[5 * (size_t)s + j + 1]) != hipSuccess) rc = fail(c, G2048_ERR_HIP, "event timing failed");
            t[j] += ms;
        }
    for (auto& ev : e)
        if (ev) (void)hipEventDestroy(ev);
    if (rc) return rc;
    for (int j = 0; j < 4; ++j) out4[j] = nsteps ? (float)(t[j] / nsteps) : 0.0f;
    return launched(c, "k_td_play/k_td_update");
}

int g2048_td_steps_profiled(g2048_ctx* c, float alpha, uint32_t nsteps, float* ms_play, float* ms_update) {
    if (!c || !ms_play || !ms_update) return c ? fail(c, G2048_ERR_ARG, "null buffer") : G2048_ERR_ARG;
    float t[4];
    if (int rc = g2048_td_steps_kernel_ms(c, alpha, nsteps, t)) return rc;
    *ms_play = t[0];
    *ms_update = t[1] + t[2] + t[3];
    return G2048_OK;
}



int g2048_debug_owner_plan(g2048_ctx* c, uint64_t* out, uint32_t capacity, uint32_t* count) {
    if (!c || !out || !count) return c ? fail(c, G2048_ERR_ARG, "null buffer") : G2048_ERR_ARG;
    NEED_TABLE(c);
    if (int rc = bind(c)) return rc;
    const uint32_t n = c->n_slices < capacity ? c->n_slices : capacity;
    std::vector<uint64_t> clk(2 * (size_t)c->n_slices);
    if (!clk.empty())
        if (int rc = d2h(c, clk.data(), c->wg_clock, clk.size() * 8)) return rc;
    for (uint32_t i = 0; i < n; ++i) {
        const Slice& s = c->plan[i];
        out[6 * i + 0] = s.variant;
        out[6 * i + 1] = s.chunk;
        out[6 * i + 2] = s.part;
        out[6 * i + 3] = s.nparts;
        out[6 * i + 4] = clk[2 * i];
        out[6 * i + 5] = clk[2 * i + 1];
    }
    *count = n;
    return G2048_OK;
}

int g2048_get_last_move(g2048_ctx* c, uint16_t* out) {
    if (!c || !out) return c ? fail(c, G2048_ERR_ARG, "null buffer") : G2048_ERR_ARG;
    if (int rc = bind(c)) return rc;
    NEED_IDENTITY(c);
    return d2h(c, out, c->last_move, (size_t)c->B * 2);
}

int g2048_log_enable(g2048_ctx* c, uint32_t lanes, uint32_t capacity) {
    if (!c) return G2048_ERR_ARG;
    NEED(c, lanes <= c->B && capacity <= (1u << 20) && (lanes == 0) == (capacity == 0), "bad log geometry");
    if (int rc = bind(c)) return rc;
    NEED_IDENTITY(c);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->log.moves) (void)hipFree(c->log.moves);
    if (c->log.start) (void)hipFree(c->log.start);
    if (c->log.final) (void)hipFree(c->log.final);
    if (c->log.meta) (void)hipFree(c->log.meta);
    c->log = GameLog{0, 0, nullptr, nullptr, nullptr, nullptr};
    if (lanes == 0) return G2048_OK;
    int rc;
    if ((rc = dalloc(c, &c->log.moves, (size_t)lanes * 2 * capacity)) || (rc = dalloc(c, &c->log.start, (size_t)lanes * 2)) ||
        (rc = dalloc(c, &c->log.final, (size_t)lanes * 2)) || (rc = dalloc(c, &c->log.meta, (size_t)lanes * 8)))
        return rc;
    c->log.lanes = lanes;
    c->log.capacity = capacity;
    k_log_init<<<grid_for(lanes), WG, 0, c->stream>>>(c->log, c->boards, c->flags);
    return launched(c, "k_log_init");
}

int g2048_log_meta(g2048_ctx* c, uint32_t* out) {
    if (!c || !out) return c ? fail(c, G2048_ERR_ARG, "null buffer") : G2048_ERR_ARG;
    if (!c->log.lanes) return fail(c, G2048_ERR_STATE, "game log is not enabled");
    if (int rc = bind(c)) return rc;
    return d2h(c, out, c->log.meta, (size_t)c->log.lanes * 8 * 4);
}

int g2048_log_game(g2048_ctx* c, uint32_t lane, uint32_t slot, uint16_t* moves, uint8_t* start) {
    if (!c || !moves || !start) return c ? fail(c, G2048_ERR_ARG, "null buffer") : G2048_ERR_ARG;
    if (!c->log.lanes) return fail(c, G2048_ERR_STATE, "game log is not enabled");
    NEED(c, lane < c->log.lanes && slot < 2, "bad lane / slot");
    if (int rc = bind(c)) return rc;
    int rc;
    if ((rc = d2h(c, moves, c->log.moves + ((size_t)lane * 2 + slot) * c->log.capacity, (size_t)c->log.capacity * 2))) return rc;
    return d2h(c, start, c->log.start + (size_t)lane * 2 + slot, 16);
}

int g2048_log_final(g2048_ctx* c, uint32_t lane, uint32_t slot, uint8_t* board) {
    if (!c || !board) return c ? fail(c, G2048_ERR_ARG, "null buffer") : G2048_ERR_ARG;
    if (!c->log.lanes) return fail(c, G2048_ERR_STATE, "game log is not enabled");
    NEED(c, lane < c->log.lanes && slot < 2, "bad lane / slot");
    if (int rc = bind(c)) return rc;
    return d2h(c, board, c->log.final + (size_t)lane * 2 + slot, 16);
}

int g2048_stats_get(g2048_ctx* c, g2048_stats* out) {
    if (!c || !out) return G2048_ERR_ARG;
    if (int rc = bind(c)) return rc;
    return d2h(c, out, c->stats, sizeof(Stats));
}

int g2048_stats_reset(g2048_ctx* c) {
    if (!c) return G2048_ERR_ARG;
    if (int rc = bind(c)) return rc;
    HIP_TRY(c, hipMemsetAsync(c->stats, 0, sizeof(Stats), c->stream));
    return G2048_OK;
}

int g2048_weights_device_ptr(g2048_ctx* c, void** ptr, int64_t* count) {
    if (!c || !ptr) return G2048_ERR_ARG;
    NEED_TABLE(c);
    *ptr = c->w;
    if (count) *count = (int64_t)c->slots;
    return G2048_OK;
}

constexpr size_t DELTA_SLACK = 256 * 64;       // floats: padding of the payload to nranks chunks of a multiple of 256 (up to 64 ranks)
static int ensure_delta(g2048_ctx* c) {
    int rc;
    if (!c->w0 && (rc = dalloc(c, &c->w0, c->slots))) return rc;
    if (!c->delta) {
        // (a zero tail behind the table's slots: the reduce-scatter form of the exchange reads whole, padded chunks — DELTA_SLACK)
        if ((rc = dalloc(c, &c->delta, c->slots + DELTA_SLACK))) return rc;
        HIP_TRY(c, hipMemsetAsync(c->delta, 0, (c->slots + DELTA_SLACK) * 4, c->stream));
    }
    return G2048_OK;
}

static int ensure_pack(g2048_ctx* c, size_t count) {
    if (c->pack && c->pack_count >= count) return G2048_OK;
    if (c->pack) HIP_TRY(c, hipFree(c->pack));
    c->pack = nullptr;
    c->pack_count = 0;
    if (int rc = dalloc(c, &c->pack, count)) return rc;
    c->pack_count = count;
    return G2048_OK;
}

// The epoch's delta D = W - W0, formed when somebody asks for it (one pass over the table per epoch).  Round 2 first kept
// D as an accumulator that every add of every step was mirrored into (G2048_DELTA_ACCUM=1 still does): two scattered
// read-modify-writes instead of one in the apply kernels, 5 % of a step at n = 5 — which a job of N > 1 ranks paid and the
// one-rank run it is compared with did not.  Both give the same D up to the rounding W itself has taken.
static int refresh_delta(g2048_ctx* c) {
    if (c->knob.delta_accum) return G2048_OK;
    k_delta_sub<<<2048, WG, 0, c->stream>>>(c->w, c->w0, c->delta, c->slots);
    return launched(c, "k_delta_sub");
}

int g2048_delta_begin(g2048_ctx* c) {
    if (!c) return G2048_ERR_ARG;
    NEED_TABLE(c);
    if (c->parent) return fail(c, G2048_ERR_STATE, "delta tracking belongs to the context that owns the table");
    if (int rc = bind(c)) return rc;
    USE_TABLE(c);
    if (int rc = ensure_delta(c)) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->w0, c->w, c->slots * 4, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->delta, 0, c->slots * 4, c->stream));
    c->tracking = true;
    return G2048_OK;
}

int g2048_delta_extract(g2048_ctx* c, void* dst) {
    if (!c) return G2048_ERR_ARG;
    NEED_TABLE(c);
    if (!c->tracking) return fail(c, G2048_ERR_STATE, "g2048_delta_begin was not called");
    if (int rc = bind(c)) return rc;
    USE_TABLE(c);
    if (int rc = refresh_delta(c)) return rc;
    if (dst) {
        k_delta_out<<<2048, WG, 0, c->stream>>>(c->delta, (float*)dst, c->slots, c->placed);
        if (int rc = launched(c, "k_delta_out")) return rc;
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return G2048_OK;
}

int g2048_delta_apply(g2048_ctx* c, const void* src) {
    if (!c) return G2048_ERR_ARG;
    NEED_TABLE(c);
    if (!c->tracking) return fail(c, G2048_ERR_STATE, "g2048_delta_begin was not called");
    if (int rc = bind(c)) return rc;
    USE_TABLE(c);
    if (!src)
        if (int rc = refresh_delta(c)) return rc;
    k_delta_add<<<2048, WG, 0, c->stream>>>(c->w, c->w0, src ? (const float*)src : c->delta, c->knob.delta_accum ? c->delta : nullptr, c->slots, src ? c->placed : 0);
    if (int rc = launched(c, "k_delta_add")) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return G2048_OK;
}

int g2048_delta_apply_mean(g2048_ctx* c, const void* pack) {
    if (!c || !pack) return c ? fail(c, G2048_ERR_ARG, "null buffer") : G2048_ERR_ARG;
    NEED_TABLE(c);
    if (!c->tracking) return fail(c, G2048_ERR_STATE, "g2048_delta_begin was not called");
    if (int rc = bind(c)) return rc;
    USE_TABLE(c);
    k_delta_add_mean<<<2048, WG, 0, c->stream>>>(c->w, c->w0, (const float*)pack, c->knob.delta_accum ? c->delta : nullptr, c->slots, c->placed);
    if (int rc = launched(c, "k_delta_add_mean")) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return G2048_OK;
}

int g2048_delta_pack_touched(g2048_ctx* c, void* pack) {
    if (!c || !pack) return c ? fail(c, G2048_ERR_ARG, "null buffer") : G2048_ERR_ARG;
    NEED_TABLE(c);
    if (!c->tracking) return fail(c, G2048_ERR_STATE, "g2048_delta_begin was not called");
    if (int rc = bind(c)) return rc;
    USE_TABLE(c);
    if (int rc = refresh_delta(c)) return rc;
    k_delta_pack_touched<<<2048, WG, 0, c->stream>>>(c->delta, (float*)pack, c->slots, c->placed);
    if (int rc = launched(c, "k_delta_pack_touched")) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return G2048_OK;
}

int g2048_delta_device_ptr(g2048_ctx* c, void** ptr) {
    if (!c || !ptr) return G2048_ERR_ARG;
    NEED_TABLE(c);
    if (int rc = bind(c)) return rc;
    if (int rc = ensure_delta(c)) return rc;
    *ptr = c->delta;
    return G2048_OK;
}

// ---- RCCL, bound at run time: the library has no link-time dependency on librccl (a process that already carries one —
// PyTorch ships its own copy under the same SONAME — gets that one back from dlopen).
namespace {
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;      // optional
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;                        // optional
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;          // optional: what the communicator itself reports
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    std::string err;
};
Rccl* rccl() {
    static Rccl r;
    if (r.lib || !r.err.empty()) return &r;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names)
        if ((r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!r.lib) {
        r.err = std::string("librccl not found: ") + dlerror();
        return &r;
    }
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.lib, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.lib, "ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(r.lib, "ncclAllReduce"));
    r.ReduceScatter = reinterpret_cast<decltype(r.ReduceScatter)>(dlsym(r.lib, "ncclReduceScatter"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.lib, "ncclAllGather"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
    r.CommCount = reinterpret_cast<decltype(r.CommCount)>(dlsym(r.lib, "ncclCommCount"));
    r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(dlsym(r.lib, "ncclCommUserRank"));
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce || !r.GetErrorString) {
        r.err = "librccl lacks an expected symbol";
        r.lib = nullptr;
    }
    return &r;
}
int rccl_fail(g2048_ctx* c, const char* what, ncclResult_t e) {
    char buf[512];
    snprintf(buf, sizeof buf, "%s: %s", what, rccl()->GetErrorString ? rccl()->GetErrorString(e) : "RCCL error");
    if (c) c->err = buf;
    return G2048_ERR_COMM;
}
}  // namespace

int g2048_comm_unique_id(uint8_t* id) {
    if (!id) return G2048_ERR_ARG;
    Rccl* r = rccl();
    if (!r->lib) return G2048_ERR_COMM;
    static_assert(sizeof(ncclUniqueId) == G2048_COMM_ID_BYTES, "unique id size");
    ncclUniqueId u;
    if (r->GetUniqueId(&u) != ncclSuccess) return G2048_ERR_COMM;
    memcpy(id, &u, sizeof u);
    return G2048_OK;
}

int g2048_comm_init(g2048_ctx* c, int rank, int nranks, const uint8_t* id) {
    if (!c || !id || nranks < 1 || rank < 0 || rank >= nranks) return c ? fail(c, G2048_ERR_ARG, "bad rank / nranks / id") : G2048_ERR_ARG;
    NEED_TABLE(c);
    if (c->parent) return fail(c, G2048_ERR_STATE, "the communicator belongs to the context that owns the table");
    if (c->comm) return fail(c, G2048_ERR_STATE, "communicator already initialised");
    if (int rc = bind(c)) return rc;
    Rccl* r = rccl();
    if (!r->lib) return fail(c, G2048_ERR_COMM, r->err.c_str());
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    ncclComm_t comm = nullptr;
    ncclResult_t e = r->CommInitRank(&comm, nranks, u, rank);       // collective: every rank calls it with the same id
    if (e != ncclSuccess) return rccl_fail(c, "ncclCommInitRank", e);
    c->comm = comm;
    c->comm_rank = rank;
    c->comm_ranks = nranks;
    return G2048_OK;
}

// What the communicator itself says (ncclCommCount / ncclCommUserRank), not what g2048_comm_init was told: the record of
// a multi-GPU run can then show that RCCL saw all N ranks.  No communicator: rank 0 of 1, G2048_OK.
int g2048_comm_info(g2048_ctx* c, int* rank, int* nranks) {
    if (!c || !rank || !nranks) return c ? fail(c, G2048_ERR_ARG, "null output") : G2048_ERR_ARG;
    *rank = 0;
    *nranks = 1;
    if (!c->comm) return G2048_OK;
    Rccl* r = rccl();
    *rank = c->comm_rank;
    *nranks = c->comm_ranks;
    if (r->CommCount && r->CommUserRank) {
        ncclResult_t e = r->CommCount((ncclComm_t)c->comm, nranks);
        if (e == ncclSuccess) e = r->CommUserRank((ncclComm_t)c->comm, rank);
        if (e != ncclSuccess) return rccl_fail(c, "ncclCommCount", e);
    }
    return G2048_OK;
}

int g2048_comm_destroy(g2048_ctx* c) {
    if (!c) return G2048_ERR_ARG;
    if (!c->comm) return G2048_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    ncclResult_t e = rccl()->CommDestroy((ncclComm_t)c->comm);
    c->comm = nullptr;
    c->comm_ranks = 1;
    return e == ncclSuccess ? G2048_OK : rccl_fail(c, "ncclCommDestroy", e);
}

// End of an epoch, all on the context's stream and without a host wait: the accumulated delta of every rank is
// sum-all-reduced over xGMI (ncclAllReduce, fp32, table_slots elements; the per-slot mean rule sends [delta | touched],
// twice that) and W = W0 + result becomes the next epoch's W0.
int g2048_allreduce_deltas(g2048_ctx* c) {
    if (!c) return G2048_ERR_ARG;
    NEED_TABLE(c);
    if (!c->comm) return fail(c, G2048_ERR_STATE, "g2048_comm_init was not called");
    if (!c->tracking) return fail(c, G2048_ERR_STATE, "g2048_delta_begin was not called");
    if (int rc = bind(c)) return rc;
    USE_TABLE(c);
    Rccl* r = rccl();
    const size_t n = c->slots;
    if (int rc = refresh_delta(c)) return rc;
    // The sum over the ranks of `count` floats of `src`, into `dst`.  Default: one ncclAllReduce.  G2048_COMM_ALGO=rsag: the same sum
    // as ncclReduceScatter + ncclAllGather (every rank reduces 1 / nranks of the payload and hands it round): on MI355X's
    // point-to-point xGMI (7 links per GPU) a ring all-reduce is bound by ONE link (2 (N-1)/N S / 153 GB/s: 4.4 ms for the n = 6
    // table at N = 8) where the direct pair keeps all seven busy (2 S / N per link: 0.63 ms) — which of the two RCCL's own
    // ncclAllReduce picks on a given box is for the 8-GPU run to show (the line's `comm` block times the exchange); both forms
    // give the same sums up to fp32 reduction order.
    auto exchange = [&](const float* src, float* dst, size_t count) -> int {      // (src may be dst; both have DELTA_SLACK floats of room, src's tail is zero)
        if (c->knob.comm_rsag && r->ReduceScatter && r->AllGather && c->comm_ranks <= 64) {
            const size_t ranks = (size_t)c->comm_ranks, chunk = ((count + ranks - 1) / ranks + 255) / 256 * 256;
            float* mine = dst + (size_t)c->comm_rank * chunk;
            ncclResult_t e = r->ReduceScatter(src, mine, chunk, ncclFloat32, ncclSum, (ncclComm_t)c->comm, c->stream);
            if (e != ncclSuccess) return rccl_fail(c, "ncclReduceScatter", e);
            e = r->AllGather(mine, dst, chunk, ncclFloat32, (ncclComm_t)c->comm, c->stream);
            if (e != ncclSuccess) return rccl_fail(c, "ncclAllGather", e);
            return G2048_OK;
        }
        const ncclResult_t e = r->AllReduce(src, dst, count, ncclFloat32, ncclSum, (ncclComm_t)c->comm, c->stream);
        return e == ncclSuccess ? G2048_OK : rccl_fail(c, "ncclAllReduce", e);
    };
    if (c->update_rule == 1) {
        if (int rc = ensure_pack(c, 2 * n + DELTA_SLACK)) return rc;
        k_delta_pack_touched<<<2048, WG, 0, c->stream>>>(c->delta, c->pack, n, 0);
        if (c->knob.comm_rsag) HIP_TRY(c, hipMemsetAsync(c->pack + 2 * n, 0, DELTA_SLACK * sizeof(float), c->stream));
        if (int rc = exchange(c->pack, c->pack, 2 * n)) return rc;
        k_delta_add_mean<<<2048, WG, 0, c->stream>>>(c->w, c->w0, c->pack, c->knob.delta_accum ? c->delta : nullptr, n, 0);
    } else {
        if (int rc = ensure_pack(c, n + DELTA_SLACK)) return rc;
        if (int rc = exchange(c->delta, c->pack, n)) return rc;
        k_delta_add<<<2048, WG, 0, c->stream>>>(c->w, c->w0, c->pack, c->knob.delta_accum ? c->delta : nullptr, n, 0);
    }
    return launched(c, "g2048_allreduce_deltas");
}

// sum of `count` doubles over the ranks, in place (episode statistics, timing): a tiny second all-reduce
int g2048_allreduce_f64(g2048_ctx* c, double* host_values, int count, int op_max) {
    if (!c || !host_values || count < 0 || count > 4096) return c ? fail(c, G2048_ERR_ARG, "bad count") : G2048_ERR_ARG;
    if (!c->comm) return fail(c, G2048_ERR_STATE, "g2048_comm_init was not called");
    if (count == 0) return G2048_OK;
    if (int rc = bind(c)) return rc;
    if (int rc = ensure_scratch(c, (size_t)count * 8)) return rc;
    if (int rc = h2d(c, c->scratch, host_values, (size_t)count * 8)) return rc;
    ncclResult_t e = rccl()->AllReduce(c->scratch, c->scratch, (size_t)count, ncclFloat64, op_max ? ncclMax : ncclSum, (ncclComm_t)c->comm, c->stream);
    if (e != ncclSuccess) return rccl_fail(c, "ncclAllReduce(f64)", e);
    return d2h(c, host_values, c->scratch, (size_t)count * 8);
}

int g2048_stream_handle(g2048_ctx* c, void** s) {
    if (!c || !s) return G2048_ERR_ARG;
    *s = (void*)c->stream;
    return G2048_OK;
}

}  // extern "C"
