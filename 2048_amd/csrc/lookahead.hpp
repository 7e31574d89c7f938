// lookahead.hpp — launch helpers of lookahead.hip (sampled expectimax on the device), called by the C ABI in g2048.hip.
#pragma once

#include "lane_state.hpp"

namespace g2048 {

struct LaPlan {
    int n;              // n-tuple size of the table (2 .. 6)
    int depth, width, since_empty;      // Game.look_forward's arguments (game_logic.py:214)
    int limit_tile;     // Game.trial_run's (game_logic.py:177): a game stops once a tile >= this is on the board (0: never)
};

constexpr int LA_MAX_DEPTH = 6, LA_MAX_WIDTH = 16;
constexpr uint64_t LA_MAX_NODES_PER_ROOT = 1ull << 22;      // deepest level of one root's tree: (4 width)^depth
constexpr uint64_t LA_NODE_BUDGET = 1ull << 26;             // nodes of all levels held at once (x 22 bytes): more roots go in rounds

// nodes of the deepest level under one root, 0 if the tree is out of range
uint64_t la_leaves_per_root(int depth, int width);
// bytes of workspace for `roots` root nodes
size_t la_workspace_bytes(uint64_t roots, int depth, int width);
// how many roots one round takes under LA_NODE_BUDGET (>= 1)
uint64_t la_roots_per_round(int depth, int width);

// V_depth of `count` boards (device pointers; salt: one (s0, s1) pair per board or null = (0, 0)); queued on `st`
hipError_t la_values(hipStream_t st, const LaPlan& plan, const float* w, const uint4* boards, const ulonglong2* salt, uint64_t count, void* ws,
                     size_t ws_bytes, float* out);
// nsteps x { Game._find_best_move with look-ahead + Game._move_on } for every live lane; queued on `st`
hipError_t la_steps(hipStream_t st, const LaPlan& plan, const float* w, LaneSet lanes, uint32_t B, int auto_reset, Stats* stats, GameLog lg, void* ws,
                    size_t ws_bytes, uint32_t nsteps);

}  // namespace g2048
