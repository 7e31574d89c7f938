// play.hpp — launch helpers of play.hip: the greedy-choice family (choose / choose_hot / choose_small and the kernels built on
// them: k_td_play*, k_eval_select*), called by the C ABI in g2048.hip.  One translation unit of its own: these are the kernels
// that take the compiler longest (the LDS-hot-set forms with their asm register contract), and nothing else changes them.
#pragma once

#include "td_types.hpp"

namespace g2048 {

constexpr int PLAY_HOT_WG = 512;        // workgroup of the LDS forms: one per CU shares the table copy (two waves per SIMD, as without it)
constexpr int PLAY_WG = 256;            // ... of the plain form

// resident workgroups per CU of the k_td_play form for (n, hot), from the occupancy calculator (0: unknown)
int play_blocks_per_cu(int n, bool hot);
// one k_td_play launch: n-tuple size, LDS form or not, `perm` non-null = the step re-orders the lanes
hipError_t launch_td_play(hipStream_t st, int n, bool hot, unsigned grid, LaneSet in, LaneSet out, const uint32_t* perm, uint4* prev_nxt, uint32_t B,
                          const float* w, float alpha, const TdRecs& recs, int auto_reset, Stats* stats, const GameLog& lg, uint32_t static_rounds);
// greedy choice of every lane (g2048_eval_select); lds: the LDS forms of n = 2, 3 (persistent grid of `grid` workgroups of PLAY_HOT_WG)
hipError_t launch_eval_select(hipStream_t st, int n, bool lds, unsigned grid, const uint4* boards, uint32_t B, const float* w, float* value,
                              uint8_t* action, float4* values4);

}  // namespace g2048
