// td_types.hpp — the records one TD step hands from k_td_play (play.hip) to the update kernels (g2048.hip), shared by both TUs.
#pragma once

#include "lane_state.hpp"

namespace g2048 {

// The (state, dw) records one TD step produces — the arguments of the calls to QAgent.update in QAgent.episode:
//   main record of lane i : state = prev[cur][i] (the previous afterstate), dw1[i] = (reward + V(after) - old_label) * alpha / F
//                           (r_learning.py:240); dw1[i] == 0 means "no record" (first move of a game, finished lane);
//   terminal records      : (this step's afterstate, -V(after) * alpha / F) for lanes whose game ended after the spawn
//                           (r_learning.py:248) — rare, so they go to a compact queue (one atomic counter bump per wave).

// k_td_play hands the last part of a launch's lane blocks out through counters.  Returning atomics on ONE address complete
// one after the other (~50 ns each on MI355X: 256 waves asking one counter cost the launch ~12 us per round), so there are
// 64 counters, each in its own cache line; the workgroups that share one sit on the same XCD (blockIdx % 64 fixes blockIdx % 8).
constexpr uint32_t PLAY_SEGS = 64, PLAY_SEG_STRIDE = 32;

struct TdRecs {
    const uint4* state1;    // prev[cur]
    float* dw1;             // [B]
    uint4* qstate;          // [B] queue of terminal-record states (packed)
    float* qdw;             // [B]
    uint32_t* qcount;       // length of this step's queue
    uint32_t* qcount_next;  // next step's counter, zeroed by k_td_play
    uint32_t* blocks;       // [PLAY_SEGS x PLAY_SEG_STRIDE] k_td_play's work counters (one per segment, a cache line each): the next 64-lane block to hand out
    uint32_t* blocks_next;  // next step's, zeroed by k_td_play
    uint32_t unit;          // 1: every record counts as dw = 1 (the counting pass of the per-slot mean rule)
    uint32_t* dwmax;        // float bits of the largest |dw| among this step's records (scale of the fixed-point sums)
    uint32_t* dwmax_next;   // next step's, zeroed by k_td_play
    // n >= 4: the orbit indices of state1, written by k_td_play when the state was chosen (OrbitIdx below); null otherwise
    const uint8_t* oidx;    // of this step's records
    uint8_t* oidx_nxt;      // of the states chosen in this step (next step's records)
};

// Orbit indices of a record's state (n >= 4).  The LDS-owner workgroups of one orbit only need that orbit's table indices
// of the images they visit (COSET_MASK: 4 per record, 1 for the centre square), not the board: k_td_play computes the 21
// of them once per lane — where VALU is idle behind the table gathers — and every one of the ~28 chunk scans reads 8 + 4
// bytes per record (indices + dw) instead of re-deriving them from the 16-byte packed state.  Orbit-major, so that a
// scan is one contiguous stream:  [4][B] uint2 (four 16-bit indices: outer line, inner line, corner square, edge square)
// | [B] uint4 (four 20-bit cross indices) | [B] uint16 (centre square).
constexpr size_t OIDX_BYTES_PER_LANE = 4 * 8 + 16 + 2;
struct OrbitIdx {
    uint2* q;       // [4][B]
    uint4* x;       // [B]
    uint16_t* c;    // [B]
};
__host__ __device__ __forceinline__ OrbitIdx orbit_idx(const uint8_t* base, size_t B) {
    uint8_t* b = const_cast<uint8_t*>(base);
    return OrbitIdx{reinterpret_cast<uint2*>(b), reinterpret_cast<uint4*>(b + 32 * B), reinterpret_cast<uint16_t*>(b + 48 * B)};
}


}  // namespace g2048
