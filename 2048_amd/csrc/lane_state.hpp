// lane_state.hpp — what every translation unit of lib2048_hip.so shares on the DEVICE side: how a lane's state is laid out and
// loaded / stored (struct-of-arrays, see g2048.hip), the episode statistics (per-workgroup LDS aggregation), the four moves of a
// board, one table gather and the value of a board, the per-lane game records.  Included inside the TU's anonymous namespace
// (everything here is a type or an inline device function in namespace g2048: the TUs share the TYPES — they appear in the
// launch helpers' signatures — and each gets its own copy of the code; no device code is linked across TUs).
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/g2048.h"
#include "features.hpp"

namespace g2048 {

constexpr int WG = 256;
constexpr uint8_t HAS_PREV = G2048_LANE_HAS_PREV;
constexpr uint8_t DONE = G2048_LANE_DONE;

__device__ __forceinline__ Board ld_board(const uint4* p, size_t i) {
    uint4 v = p[i];
    Board b;
    b.r[0] = v.x; b.r[1] = v.y; b.r[2] = v.z; b.r[3] = v.w;
    return b;
}
#define G2048_ST(ptr, val) (*(ptr) = (val))
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st_board(uint4* p, size_t i, const Board& b) {
    u32x4_t v = {b.r[0], b.r[1], b.r[2], b.r[3]};
    G2048_ST(reinterpret_cast<u32x4_t*>(p) + i, v);
}
__device__ __forceinline__ Rng ld_rng(const ulonglong2* p, size_t i) {
    ulonglong2 v = p[i];
    Rng g;
    g.s0 = v.x; g.s1 = v.y;
    return g;
}
__device__ __forceinline__ void st_rng(ulonglong2* p, size_t i, const Rng& g) {
    u64x2_t v = {g.s0, g.s1};
    G2048_ST(reinterpret_cast<u64x2_t*>(p) + i, v);
}

// `state` records are kept in packed form (features.hpp): x = R0|R1<<16, y = R2|R3<<16, z = C0|C1<<16, w = C2|C3<<16
__device__ __forceinline__ Packed ld_packed(const uint4* p, size_t i) {
    uint4 v = p[i];
    Packed q;
    q.R[0] = v.x & 0xFFFFu; q.R[1] = v.x >> 16; q.R[2] = v.y & 0xFFFFu; q.R[3] = v.y >> 16;
    q.C[0] = v.z & 0xFFFFu; q.C[1] = v.z >> 16; q.C[2] = v.w & 0xFFFFu; q.C[3] = v.w >> 16;
    return q;
}
__device__ __forceinline__ void st_packed(uint4* p, size_t i, const Packed& q) {
    u32x4_t v = {q.R[0] | (q.R[1] << 16), q.R[2] | (q.R[3] << 16), q.C[0] | (q.C[1] << 16), q.C[2] | (q.C[3] << 16)};
    G2048_ST(reinterpret_cast<u32x4_t*>(p) + i, v);
}

struct Stats {   // device mirror of g2048_stats (all u64)
    unsigned long long episodes, moves, score_sum, best_score, max_tile[20], overflow16, nonfinite, valid_dirs;
};
static_assert(sizeof(Stats) == sizeof(g2048_stats), "stats layout");

// Episode statistics are accumulated per workgroup in LDS and flushed with one global atomic per non-zero counter:
// thousands of lanes finishing in the same step would otherwise all hit the same few addresses (same-address global
// atomics run at well under 1 G/s).
struct WgStats {
    unsigned long long score_sum;
    unsigned int episodes, moves, valid_dirs, best, overflow16, nonfinite, max_tile[20];
    unsigned int dw_max_bits;       // largest |dw| of the workgroup's records, as float bits (orders like an unsigned int)
};

__device__ __forceinline__ void wg_stats_init(WgStats* ws) {
    if (threadIdx.x < 20) ws->max_tile[threadIdx.x] = 0;
    if (threadIdx.x == 0) {
        ws->score_sum = 0;
        ws->episodes = ws->moves = ws->valid_dirs = ws->best = ws->overflow16 = ws->nonfinite = 0;
        ws->dw_max_bits = 0;
    }
    __syncthreads();
}

__device__ __forceinline__ void count_finished(WgStats* ws, const Board& b, int32_t score, bool overflow) {
    const unsigned int sc = score < 0 ? 0u : (unsigned int)score;
    atomicAdd(&ws->episodes, 1u);
    atomicAdd(&ws->score_sum, (unsigned long long)sc);
    atomicMax(&ws->best, sc);
    uint32_t t = max_tile(b);
    atomicAdd(&ws->max_tile[t > 19u ? 19u : t], 1u);
    if (overflow) atomicAdd(&ws->overflow16, 1u);
}

// board-steps executed and directions that were open to them: one LDS add per wave each
__device__ __forceinline__ void count_moves(WgStats* ws, unsigned int my_moves, unsigned int my_dirs) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        my_moves += __shfl_down(my_moves, off);
        my_dirs += __shfl_down(my_dirs, off);
    }
    if ((threadIdx.x & 63) == 0 && my_moves) {
        atomicAdd(&ws->moves, my_moves);
        atomicAdd(&ws->valid_dirs, my_dirs);
    }
}

__device__ __forceinline__ void wg_stats_flush(const WgStats* ws, Stats* st) {
    __syncthreads();
    if (threadIdx.x < 20 && ws->max_tile[threadIdx.x]) atomicAdd(&st->max_tile[threadIdx.x], (unsigned long long)ws->max_tile[threadIdx.x]);
    if (threadIdx.x == 32) {
        if (ws->episodes) atomicAdd(&st->episodes, (unsigned long long)ws->episodes);
        if (ws->moves) atomicAdd(&st->moves, (unsigned long long)ws->moves);
        if (ws->valid_dirs) atomicAdd(&st->valid_dirs, (unsigned long long)ws->valid_dirs);
        if (ws->score_sum) atomicAdd(&st->score_sum, ws->score_sum);
        if (ws->best) atomicMax(&st->best_score, (unsigned long long)ws->best);
        if (ws->overflow16) atomicAdd(&st->overflow16, (unsigned long long)ws->overflow16);
        if (ws->nonfinite) atomicAdd(&st->nonfinite, (unsigned long long)ws->nonfinite);
    }
}


struct Moves4 {          // four named members, not an array: keeps every field a scalar the compiler can hold in a register
    Moved m0, m1, m2, m3;
};

__device__ __forceinline__ Moves4 all_moves(const Board& b) {
    uint32_t cols[4];
    transpose(b.r, cols);
    Moves4 r;
    r.m0 = move_dir<0>(b.r, cols);
    r.m1 = move_dir<1>(b.r, cols);
    r.m2 = move_dir<2>(b.r, cols);
    r.m3 = move_dir<3>(b.r, cols);
    return r;
}

__device__ __forceinline__ uint8_t changed_mask(const Moves4& mv) {
    return (uint8_t)((mv.m0.changed ? 1u : 0u) | (mv.m1.changed ? 2u : 0u) | (mv.m2.changed ? 4u : 0u) | (mv.m3.changed ? 8u : 0u));
}

// The move of direction d (runtime), blended with bit masks.  A `?:` chain here would be folded by the compiler into a
// load through a selected stack address, which parks all four moves in scratch memory (116 B per lane).
__device__ __forceinline__ Moved pick(const Moves4& mv, uint32_t d) {
    const uint32_t k0 = 0u - (uint32_t)(d == 0), k1 = 0u - (uint32_t)(d == 1), k2 = 0u - (uint32_t)(d == 2), k3 = 0u - (uint32_t)(d == 3);
#define G2048_BLEND(field) ((mv.m0.field & k0) | (mv.m1.field & k1) | (mv.m2.field & k2) | (mv.m3.field & k3))
    Moved o;
    o.after.r[0] = G2048_BLEND(after.r[0]);
    o.after.r[1] = G2048_BLEND(after.r[1]);
    o.after.r[2] = G2048_BLEND(after.r[2]);
    o.after.r[3] = G2048_BLEND(after.r[3]);
    o.ma = G2048_BLEND(ma);
    o.mb = G2048_BLEND(mb);
    o.changed = ((uint32_t)changed_mask(mv) >> d) & 1u;
#undef G2048_BLEND
    return o;
}


// QAgent.evaluate (r_learning.py:202-203): left-to-right sum of one weight per feature (fp32 here, float64 there)
// One table entry.  (Forcing the scalar-base + 32-bit-vector-offset form of global_load — an opaque 32-bit byte offset,
// so that a pending gather holds one address register instead of a 64-bit pair — was measured: same register count after
// allocation, k_td_play 0.200 -> 0.210 ms.  Plain indexing it is.)
// (the slots handed to ld_w are memory slots: memory_slots<N>, features.hpp)
__device__ __forceinline__ float ld_w(const float* __restrict__ w, uint32_t slot) { return w[slot]; }

// the gather of feature f (a non-temporal form for the f_6 tables, whose 361 MB have next to no reuse in a CU's 32 KB L1, was
// measured: no difference — the L1 miss path carries the request either way)
template <int N>
__device__ __forceinline__ float ld_w_f(const float* __restrict__ w, uint32_t slot, int f) {
    return ld_w(w, slot);
}

template <int N>
__device__ __forceinline__ float value_of(const float* __restrict__ w, const Board& b) {
    constexpr int F = Shape<N>::F;
    uint32_t s[F];
    memory_slots<N>(pack_pairs(b), s);
    float x[F];
#pragma unroll
    for (int f = 0; f < F; ++f) x[f] = ld_w(w, s[f]);
    float v = 0.0f;
#pragma unroll
    for (int f = 0; f < F; ++f) v += x[f];
    return v;
}


// The per-lane arrays that carry a game from one step to the next.  There are two such sets: a step that re-orders the
// lanes (LaneSort below) reads one through the permutation and writes the other in the new order.
struct LaneSet {
    uint4* boards;
    int32_t* scores;
    ulonglong2* rng;
    float* label;
    uint8_t* flags;
    uint32_t* lane_id;      // which lane of the context (0 .. B-1, the index the host sees) sits at this position
    uint16_t* last_move;
};

// Optional per-lane game records for the first `lanes` lanes (Game.moves / Game.tiles / starting_position of
// game_logic.py:55-66, what Game.replay and show.py's replay need).  Two slots per lane: while one game is being
// written the previous, finished one stays readable.
struct GameLog {
    uint32_t lanes, capacity;       // lanes == 0: off
    uint16_t* moves;                // [lanes][2][capacity] g2048_get_last_move words, one per move
    uint4* start;                   // [lanes][2] starting boards
    uint4* final;                   // [lanes][2] boards the games ended on
    uint32_t* meta;                 // [lanes][8]: slot in use, moves so far, games finished, {length, score} of slot 0, of slot 1, flags
};
constexpr uint32_t LOG_PARTIAL0 = 1u, LOG_TRUNC0 = 4u;     // flags: bit s = slot s did not start at move 0; bit 2 + s = slot s overflowed


__device__ __forceinline__ void log_step(const GameLog& lg, uint32_t i, uint32_t lm, bool moved, bool over, int32_t final_score,
                                         bool restarted, const Board& fresh, const Board& last) {
    uint32_t* m = lg.meta + 8 * i;
    uint32_t slot = m[0], cnt = m[1];
    if (moved) {
        if (cnt < lg.capacity)
            lg.moves[((size_t)i * 2 + slot) * lg.capacity + cnt] = (uint16_t)lm;
        else
            m[7] |= LOG_TRUNC0 << slot;
        ++cnt;
    }
    if (over) {
        m[3 + 2 * slot] = cnt;
        m[4 + 2 * slot] = (uint32_t)final_score;
        m[2] += 1;
        st_board(lg.final, (size_t)i * 2 + slot, last);
        if (restarted) {
            slot ^= 1u;
            cnt = 0;
            m[3 + 2 * slot] = 0;
            m[7] &= ~((LOG_PARTIAL0 | LOG_TRUNC0) << slot);
            st_board(lg.start, (size_t)i * 2 + slot, fresh);
        }
    }
    m[0] = slot;
    m[1] = cnt;
}

}  // namespace g2048
