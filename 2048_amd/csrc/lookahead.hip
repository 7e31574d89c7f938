// lookahead.hip — Game.look_forward (game2048/game_logic.py:214-243), the reference's sampled expectimax, on the device.
//
//     V_0(s)        = estimator(s)
//     V_d(s), d > 0 = estimator(s)                                              if empty(s) >= since_empty
//                   = (1 / k) * sum over k = min(width, empty(s)) sampled tiles t of max(0, W(s + t))
//     W(c)          = -100                                                      if c is game over
//                   = max over the directions that change c of V_{d-1}(after_dir(c))
//
// The reference walks one tree depth-first in Python ("1 second per move" at depth 3, width 4: README.md:145).  Here the trees
// of ALL positions asked about — the four candidate moves of every game of a trial — are one implicit complete (4 width)-ary
// tree per root, held level by level in HBM and never leaving it:
//
//   level l holds roots x (4 width)^l nodes (afterstates): board 16 B | value 4 B | valid 1 B | k 1 B, struct-of-arrays;
//   node p's children sit at (p * width + j) * 4 + dir — sampled tile j, direction dir — so a level needs no index lists, no
//   compaction and no atomics: a node that does not exist (direction does not change the board, j >= k, parent is a leaf) is
//   a zero in `valid`;
//   k_la_expand   one thread per (node, j): leaf test, the j-th sampled tile, the four moves -> four child nodes (64 B store);
//                 a node that is a leaf by `since_empty` is evaluated on the spot (table gathers, value_of)
//   k_la_eval     the deepest level: every valid node is a leaf
//   k_la_reduce   bottom-up, one thread per node: mean over j of max(0, best child) — the arithmetic above in fp32,
//                 left to right, as the reference's loop adds it up
//   k_la_pick     (games) first maximum over the root's directions, the move, the new tile from the lane's own stream,
//                 terminal test, statistics, game record — what Game._find_best_move / _move_on / trial_run do around it.
//
// Chance nodes: the reference samples with Python's global Mersenne Twister in depth-first order; here a node's samples are a
// function of the node (its board) and a salt — 2048_amd/rng.py: lookahead_draws is the spec — so any visiting order gives the
// same tree, and the reference itself can be fed the same draws (tests/golden/make_golden4.py), which pins values, best moves
// and whole look-ahead games against it.  For a game the salt is the lane's RNG state at the move (no draw is consumed).
//
// All gathers go through value_of<N> (lane_state.hpp) — the table is read in its memory order like everywhere else.  No MFMA:
// the work is SWAR moves and 4-byte gathers.
#include <hip/hip_runtime.h>
#include <math.h>

#include "lookahead.hpp"

namespace g2048 {

namespace {

struct Level {
    uint4* board;
    float* val;
    uint8_t* valid;     // 0: no such node
    uint8_t* kcnt;      // (valid nodes) 0: a leaf, its value is in `val`; k > 0: inner node with k sampled tiles
    uint32_t count;
};

// the stream a node's chance samples come from (spec: rng.py lookahead_rng)
__device__ __forceinline__ Rng la_rng(const Board& b, const ulonglong2 salt) {
    const uint64_t lo = (uint64_t)b.r[0] | (uint64_t)b.r[1] << 32, hi = (uint64_t)b.r[2] | (uint64_t)b.r[3] << 32;
    uint64_t x = salt.x ^ lo;
    const uint64_t a = splitmix64(x);
    x = (x ^ hi) + salt.y;
    Rng g;
    g.s0 = splitmix64(x);
    x ^= a;
    g.s1 = splitmix64(x);
    if ((g.s0 | g.s1) == 0) g.s0 = 1;
    return g;
}

template <int N>
__global__ __launch_bounds__(WG) void k_la_expand(Level cur, Level nxt, uint32_t width, uint32_t since_empty, uint32_t per_root, const ulonglong2* salt,
                                                  const float* __restrict__ w) {
    const uint32_t t = blockIdx.x * WG + threadIdx.x;
    const uint32_t p = t / width, j = t - p * width;
    if (p >= cur.count) return;
    uint32_t* const child_valid = reinterpret_cast<uint32_t*>(nxt.valid + (size_t)t * 4u);      // the four children's bytes
    if (!cur.valid[p]) {
        *child_valid = 0u;
        if (j == 0) cur.kcnt[p] = 0;
        return;
    }
    const Board b = ld_board(cur.board, p);
    const uint32_t e = empty_bits(b), ne = popcount32(e);
    if (ne >= since_empty) {            // game_logic.py:218-219: roomy boards are not searched
        *child_valid = 0u;
        if (j == 0) {
            cur.kcnt[p] = 0;
            cur.val[p] = value_of<N>(w, b);
        }
        return;
    }
    const uint32_t k = ne < width ? ne : width;         // :220
    if (j == 0) {
        cur.kcnt[p] = (uint8_t)k;
        if (k == 0) cur.val[p] = __builtin_nanf("");    // a full board: the reference divides by zero here (:242)
    }
    if (j >= k) {
        *child_valid = 0u;
        return;
    }
    // the j-th sampled (cell, tile) of this node: draws 0 .. j of its stream, cells without replacement (rng.py: lookahead_draws)
    Rng g = la_rng(b, salt[p / per_root]);
    uint32_t free_cells = e, r10 = 0, kk = 0;
    for (uint32_t i = 0;; ++i) {
        spawn_draw(next_u64(g), popcount32(free_cells), r10, kk);
        if (i == j) break;
        free_cells &= ~(1u << kth_set_bit(free_cells, kk));
    }
    Board child = b;
    place_tile(child, r10, kk, free_cells);
    const Moves4 mv = all_moves(child);
    const size_t c = (size_t)t * 4u;
    st_board(nxt.board, c + 0, mv.m0.after);
    st_board(nxt.board, c + 1, mv.m1.after);
    st_board(nxt.board, c + 2, mv.m2.after);
    st_board(nxt.board, c + 3, mv.m3.after);
    *child_valid = (mv.m0.changed ? 1u : 0u) | (mv.m1.changed ? 1u << 8 : 0u) | (mv.m2.changed ? 1u << 16 : 0u) | (mv.m3.changed ? 1u << 24 : 0u);
}

template <int N>
__global__ __launch_bounds__(WG) void k_la_eval(Level last, const float* __restrict__ w) {
    const uint32_t p = blockIdx.x * WG + threadIdx.x;
    if (p >= last.count || !last.valid[p]) return;
    last.kcnt[p] = 0;
    last.val[p] = value_of<N>(w, ld_board(last.board, p));
}

// game_logic.py:226-242 for one inner node: `average += max(best_value, 0)` over its sampled tiles, then `/ num_tiles`
__global__ __launch_bounds__(WG) void k_la_reduce(Level cur, Level nxt, uint32_t width) {
    const uint32_t p = blockIdx.x * WG + threadIdx.x;
    if (p >= cur.count || !cur.valid[p]) return;
    const uint32_t k = cur.kcnt[p];
    if (k == 0) return;                 // a leaf: evaluated already
    float acc = 0.0f;
    for (uint32_t j = 0; j < k; ++j) {
        const size_t c = ((size_t)p * width + j) * 4u;
        const uint32_t valid4 = *reinterpret_cast<const uint32_t*>(nxt.valid + c);
        const float4 v = *reinterpret_cast<const float4*>(nxt.val + c);
        float best = -INFINITY;         // Python's max(best_value, value): the new value only if it is greater
        if ((valid4 & 0xFFu) && v.x > best) best = v.x;
        if ((valid4 & 0xFF00u) && v.y > best) best = v.y;
        if ((valid4 & 0xFF0000u) && v.z > best) best = v.z;
        if ((valid4 & 0xFF000000u) && v.w > best) best = v.w;
        const float wj = valid4 ? best : -100.0f;       // no direction changes the child: game over (:229-231)
        acc += (0.0f > wj) ? 0.0f : wj;                 // max(best_value, 0)
    }
    cur.val[p] = acc / (float)k;
}

// level 0 of a game step: the (up to) four afterstates of every live lane; the salt of its chance nodes = the lane's RNG state
__global__ __launch_bounds__(WG) void k_la_roots_lanes(LaneSet lanes, uint32_t lo, uint32_t hi, Level l0, ulonglong2* salt, uint32_t limit_tile) {
    const uint32_t r = blockIdx.x * WG + threadIdx.x, i = lo + r;
    if (i >= hi) return;
    uint32_t valid4 = 0;
    const Board b = ld_board(lanes.boards, i);
    const bool live = !(lanes.flags[i] & DONE) && !(limit_tile && max_tile(b) >= limit_tile);
    if (live) {
        const Moves4 mv = all_moves(b);
        st_board(l0.board, (size_t)r * 4 + 0, mv.m0.after);
        st_board(l0.board, (size_t)r * 4 + 1, mv.m1.after);
        st_board(l0.board, (size_t)r * 4 + 2, mv.m2.after);
        st_board(l0.board, (size_t)r * 4 + 3, mv.m3.after);
        valid4 = (mv.m0.changed ? 1u : 0u) | (mv.m1.changed ? 1u << 8 : 0u) | (mv.m2.changed ? 1u << 16 : 0u) | (mv.m3.changed ? 1u << 24 : 0u);
    }
    *reinterpret_cast<uint32_t*>(l0.valid + (size_t)r * 4) = valid4;
    const ulonglong2 s = lanes.rng[i];
#pragma unroll
    for (int d = 0; d < 4; ++d) salt[(size_t)r * 4 + d] = s;
}

// Game._find_best_move's choice (game_logic.py:150-161: first maximum, strict '>') and Game._move_on (:163-167) for every live
// lane; the loop conditions of Game.trial_run (:174-178) decide whether the lane goes on.
__global__ __launch_bounds__(WG) void k_la_pick(LaneSet lanes, uint32_t lo, uint32_t hi, Level l0, int auto_reset, Stats* stats, GameLog lg, uint32_t limit_tile) {
    __shared__ WgStats ws;
    wg_stats_init(&ws);
    const uint32_t r = blockIdx.x * WG + threadIdx.x, i = lo + r;
    uint32_t my_moves = 0, my_dirs = 0;
    if (i < hi) {
        uint8_t fl = lanes.flags[i];
        if (!(fl & DONE)) {
            Board b = ld_board(lanes.boards, i);
            Rng g = ld_rng(lanes.rng, i);
            int32_t score = lanes.scores[i];
            const uint32_t valid4 = *reinterpret_cast<const uint32_t*>(l0.valid + (size_t)r * 4);
            const float4 v = *reinterpret_cast<const float4*>(l0.val + (size_t)r * 4);
            const float vd[4] = {v.x, v.y, v.z, v.w};
            int action = -1, first_valid = -1;
            float best = -INFINITY;
#pragma unroll
            for (int d = 0; d < 4; ++d)
                if ((valid4 >> (8 * d)) & 0xFFu) {
                    if (first_valid < 0) first_valid = d;
                    if (vd[d] > best) {
                        best = vd[d];
                        action = d;
                    }
                }
            if (action < 0) action = first_valid;       // every value NaN (a poisoned table): still a legal move
            uint32_t lm = 0;
            bool moved = false, over, overflow = false;
            if (action >= 0) {
                const Moves4 mv = all_moves(b);
                const Moved ch = pick(mv, (uint32_t)action);
                score += (int32_t)merged_score(ch.ma, ch.mb);
                b = ch.after;
                moved = true;
                my_moves = 1;
                my_dirs = popcount32(changed_mask(mv));
                lm = (uint32_t)action | 4u;
                if (spawn(b, g)) {
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const uint32_t d = b.r[rr] ^ ch.after.r[rr];        // the one byte that changed
                        if (d) {
                            const uint32_t col = (uint32_t)(__ffs((int)d) - 1) >> 3;
                            lm |= ((uint32_t)(4 * rr) + col) << 4 | ((d >> (8 * col)) & 3u) << 8 | 1u << 10;
                        }
                    }
                }
                overflow = max_tile(b) >= 16u;
                over = game_over(b) || overflow || (limit_tile && max_tile(b) >= limit_tile);
            } else {
                over = true;            // a dead board, or one already at the tile limit: trial_run makes no move
            }
            const int32_t final_score = score;
            const Board final_board = b;
            if (over) {
                lm |= 1u << 11;
                count_finished(&ws, b, score, overflow);
                if (auto_reset) {
                    b = new_game(g);
                    score = 0;
                } else {
                    fl |= DONE;
                }
            }
            if (i < lg.lanes) log_step(lg, i, lm, moved, over, final_score, over && auto_reset, b, final_board);
            st_board(lanes.boards, i, b);
            st_rng(lanes.rng, i, g);
            lanes.scores[i] = score;
            lanes.flags[i] = fl;
            lanes.last_move[i] = (uint16_t)lm;
        } else {
            lanes.last_move[i] = 0;
        }
    }
    count_moves(&ws, my_moves, my_dirs);
    wg_stats_flush(&ws, stats);
}

inline unsigned grid_for(uint64_t threads) { return (unsigned)((threads + WG - 1) / WG); }
inline size_t align16(size_t x) { return (x + 15u) & ~(size_t)15u; }

struct Tree {
    Level lv[LA_MAX_DEPTH + 1];
    ulonglong2* salt;
};

// carve the workspace: salt | per level: board, val, valid, kcnt (level 0's boards may live elsewhere: `own_roots`)
Tree carve(void* ws, uint64_t roots, int depth, int width) {
    Tree t{};
    uint8_t* p = static_cast<uint8_t*>(ws);
    t.salt = reinterpret_cast<ulonglong2*>(p);
    p += align16(roots * sizeof(ulonglong2));
    uint64_t count = roots;
    for (int l = 0; l <= depth; ++l) {
        Level& L = t.lv[l];
        L.count = (uint32_t)count;
        L.board = reinterpret_cast<uint4*>(p);
        p += align16(count * 16);
        L.val = reinterpret_cast<float*>(p);
        p += align16(count * 4);
        L.valid = p;
        p += align16(count);
        L.kcnt = p;
        p += align16(count);
        count *= (uint64_t)(4 * width);
    }
    return t;
}

template <int N>
void run_tree(hipStream_t st, const Tree& t, const LaPlan& plan, const float* w) {
    uint32_t per_root = 1;
    for (int l = 0; l < plan.depth; ++l) {
        k_la_expand<N><<<grid_for((uint64_t)t.lv[l].count * plan.width), WG, 0, st>>>(t.lv[l], t.lv[l + 1], (uint32_t)plan.width, (uint32_t)plan.since_empty, per_root,
                                                                                       t.salt, w);
        per_root *= 4u * (uint32_t)plan.width;
    }
    k_la_eval<N><<<grid_for(t.lv[plan.depth].count), WG, 0, st>>>(t.lv[plan.depth], w);
    for (int l = plan.depth - 1; l >= 0; --l) k_la_reduce<<<grid_for(t.lv[l].count), WG, 0, st>>>(t.lv[l], t.lv[l + 1], (uint32_t)plan.width);
}

void run_tree_n(hipStream_t st, const Tree& t, const LaPlan& plan, const float* w) {
    switch (plan.n) {
        case 2: run_tree<2>(st, t, plan, w); break;
        case 3: run_tree<3>(st, t, plan, w); break;
        case 4: run_tree<4>(st, t, plan, w); break;
        case 5: run_tree<5>(st, t, plan, w); break;
        default: run_tree<6>(st, t, plan, w); break;
    }
}

}  // namespace

uint64_t la_leaves_per_root(int depth, int width) {
    if (depth < 0 || depth > LA_MAX_DEPTH || width < 1 || width > LA_MAX_WIDTH) return 0;
    uint64_t leaves = 1;
    for (int l = 0; l < depth; ++l) {
        leaves *= (uint64_t)(4 * width);
        if (leaves > LA_MAX_NODES_PER_ROOT) return 0;
    }
    return leaves;
}

static uint64_t nodes_per_root(int depth, int width) {
    uint64_t total = 0, level = 1;
    for (int l = 0; l <= depth; ++l) {
        total += level;
        level *= (uint64_t)(4 * width);
    }
    return total;
}

uint64_t la_roots_per_round(int depth, int width) {
    const uint64_t per = nodes_per_root(depth, width);
    const uint64_t roots = LA_NODE_BUDGET / per;
    return roots < 4 ? 4 : roots & ~(uint64_t)3;        // (a lane brings four roots)
}

size_t la_workspace_bytes(uint64_t roots, int depth, int width) {
    size_t bytes = align16(roots * sizeof(ulonglong2));
    uint64_t count = roots;
    for (int l = 0; l <= depth; ++l) {
        bytes += align16(count * 16) + align16(count * 4) + 2 * align16(count);
        count *= (uint64_t)(4 * width);
    }
    return bytes;
}

hipError_t la_values(hipStream_t st, const LaPlan& plan, const float* w, const uint4* boards, const ulonglong2* salt, uint64_t count, void* ws, size_t ws_bytes,
                     float* out) {
    const uint64_t round = la_roots_per_round(plan.depth, plan.width);
    for (uint64_t lo = 0; lo < count; lo += round) {
        const uint64_t roots = count - lo < round ? count - lo : round;
        if (la_workspace_bytes(roots, plan.depth, plan.width) > ws_bytes) return hipErrorInvalidValue;
        Tree t = carve(ws, roots, plan.depth, plan.width);
        t.lv[0].board = const_cast<uint4*>(boards) + lo;        // (level 0 is only read)
        hipError_t e = hipMemsetAsync(t.lv[0].valid, 1, roots, st);
        if (e != hipSuccess) return e;
        if (salt)
            e = hipMemcpyAsync(t.salt, salt + lo, roots * sizeof(ulonglong2), hipMemcpyDeviceToDevice, st);
        else
            e = hipMemsetAsync(t.salt, 0, roots * sizeof(ulonglong2), st);
        if (e != hipSuccess) return e;
        run_tree_n(st, t, plan, w);
        e = hipMemcpyAsync(out + lo, t.lv[0].val, roots * sizeof(float), hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return e;
    }
    return hipGetLastError();
}

hipError_t la_steps(hipStream_t st, const LaPlan& plan, const float* w, LaneSet lanes, uint32_t B, int auto_reset, Stats* stats, GameLog lg, void* ws,
                    size_t ws_bytes, uint32_t nsteps) {
    const uint32_t round = (uint32_t)(la_roots_per_round(plan.depth, plan.width) / 4);      // lanes per round
    for (uint32_t s = 0; s < nsteps; ++s)
        for (uint32_t lo = 0; lo < B; lo += round) {
            const uint32_t hi = B - lo < round ? B : lo + round;
            const uint64_t roots = (uint64_t)(hi - lo) * 4;
            if (la_workspace_bytes(roots, plan.depth, plan.width) > ws_bytes) return hipErrorInvalidValue;
            const Tree t = carve(ws, roots, plan.depth, plan.width);
            k_la_roots_lanes<<<grid_for(hi - lo), WG, 0, st>>>(lanes, lo, hi, t.lv[0], t.salt, (uint32_t)plan.limit_tile);
            run_tree_n(st, t, plan, w);
            k_la_pick<<<grid_for(hi - lo), WG, 0, st>>>(lanes, lo, hi, t.lv[0], auto_reset, stats, lg, (uint32_t)plan.limit_tile);
        }
    return hipGetLastError();
}

}  // namespace g2048
