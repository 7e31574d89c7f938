/* g2048.h — C ABI of the MI355X-native batched 2048 environment + n-tuple TD(0) learner.
 *
 * The reference (abachurin/2048) has no FFI: its hot path sits behind two Python classes, `Game`
 * (game2048/game_logic.py) and `QAgent` (game2048/r_learning.py).  Each entry point below names the
 * reference code it replaces; INTEGRATION.md shows the ctypes binding a maintainer adds on the Python side.
 *
 * Conventions
 *   - every function returns an int status: 0 = G2048_OK, negative = error (g2048_strerror / g2048_last_error);
 *     nothing throws or aborts across the boundary;
 *   - host buffers are caller-owned plain arrays, copied synchronously; sizes are implied by the context
 *     (B = batch, F = g2048_num_feat(n)) unless a count is passed;
 *   - a board is 16 uint8 log2-tiles, row-major (0 = empty, k = tile 2^k), i.e. game_logic.py:62 narrowed to u8;
 *   - directions: 0 left, 1 up, 2 right, 3 down (Game.actions, game_logic.py:50);
 *   - a context owns all device memory and one HIP stream; it is not re-entrant (the caller serialises calls
 *     on one context); distinct contexts are independent.  ctypes releases the GIL around every call.
 *   - ASYNCHRONY: calls that only launch device work (g2048_reset, g2048_step_random, g2048_weights_init,
 *     g2048_td_steps, g2048_stats_reset, g2048_clear_carry, g2048_allreduce_deltas ...) return once the work is queued on
 *     the context's stream; calls that move data to or from host buffers wait for the stream first, and g2048_sync waits
 *     explicitly.  Errors of queued kernels surface at the next synchronising call.
 *   - SINGLE WRITER: contexts made with g2048_create_shared run on their own streams over ONE weight table.  The
 *     library orders their device work by the order of the CALLS (each table-using call first makes its stream wait
 *     for the latest table-using call of any sibling), so the caller must issue calls on contexts that share a table
 *     from one thread at a time, and two such contexts never step concurrently.
 *   - TWO LIBRARIES export this ABI: lib2048_hip.so (2048_amd/csrc/g2048.hip — the product; g2048_create fails with
 *     G2048_ERR_NODEV without a GPU and never routes anywhere else) and lib2048_cpu.so (2048_amd/csrc/cpu_ref.cpp — the same
 *     entry points in scalar C++ from the kernels' integer headers: SURVEY.md 8b's `cpu_ref`).  Which one a process loads is
 *     the caller's explicit choice; neither falls back to the other.  Differences of the CPU library: "device" pointers are
 *     host pointers, device 0 is the host, g2048_comm_* / g2048_allreduce_* return G2048_ERR_COMM, the debug entry points of
 *     GPU-only machinery (lane order, owner plan) report nothing.
 */
#ifndef G2048_H
#define G2048_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define G2048_ABI_VERSION 3

enum {
    G2048_OK = 0,
    G2048_ERR_ARG = -1,    /* bad argument (null pointer, bad n-tuple, count mismatch) */
    G2048_ERR_HIP = -2,    /* a HIP runtime call failed; see g2048_last_error */
    G2048_ERR_NOMEM = -3,  /* device or host allocation failed */
    G2048_ERR_STATE = -4,  /* call not valid in this state (e.g. no weight table: n_tuple == 0) */
    G2048_ERR_NODEV = -5,  /* no usable GPU */
    G2048_ERR_COMM = -6    /* RCCL is missing or a collective failed; see g2048_last_error */
};

/* lane flag bits (g2048_get_carry) */
#define G2048_LANE_HAS_PREV 1u /* `state is not None` in QAgent.episode, r_learning.py:226,238 */
#define G2048_LANE_DONE 2u     /* episode finished and auto-reset is off */

typedef struct g2048_ctx g2048_ctx;

typedef struct g2048_stats {
    uint64_t episodes;       /* games finished                         (self.step, r_learning.py:251) */
    uint64_t moves;          /* board-steps executed                   (Game.odometer summed)         */
    uint64_t score_sum;      /* sum of final scores                    (av1000, r_learning.py:298)    */
    uint64_t best_score;     /* best final score                       (top_score, r_learning.py:302) */
    uint64_t max_tile[20];   /* histogram of the largest tile of finished games (reached[], :307-309) */
    uint64_t overflow16;     /* lanes ended because a 15+15 merge left the reference's tile domain    */
    uint64_t nonfinite;      /* TD records dropped because their dw was inf / NaN (a poisoned table)  */
    uint64_t valid_dirs;     /* directions that changed the board, summed over the board-steps (`change` true, :233) */
} g2048_stats;

/* ---- library / geometry (host only) */
int g2048_abi_version(void);
const char* g2048_strerror(int status);
int g2048_device_count(int* count);
int g2048_num_feat(int n_tuple);                      /* QAgent.parameter_shape, r_learning.py:88 -> 24/52/17/21/33 */
int64_t g2048_table_slots(int n_tuple);               /* total fp32 slots of the flat table, init_weights :136-149 */
int g2048_feature_layout(int n_tuple, int64_t* offsets, int64_t* sizes); /* per-feature first slot / slot count */

/* ---- lifecycle.  n_tuple in {0 (environment only), 2, 3, 4, 5, 6}.  Lanes are seeded from
 * splitmix64(seed + lane0 + lane) and start as fresh games (Game.__init__, game_logic.py:55-66). */
int g2048_create(int device, uint32_t batch, int n_tuple, uint64_t seed, uint64_t lane0, g2048_ctx** out);
/* A second set of lanes over the SAME weight table as `parent` (same device, same n-tuple): e.g. one lane for
 * QAgent.episode / evaluate next to the big training batch.  The parent must outlive it; the caller serialises
 * calls on contexts that share a table (SINGLE WRITER above); device work already queued on the table by the parent
 * is ordered before the child's. */
int g2048_create_shared(g2048_ctx* parent, uint32_t batch, uint64_t seed, uint64_t lane0, g2048_ctx** out);
int g2048_destroy(g2048_ctx* ctx);
const char* g2048_last_error(const g2048_ctx* ctx);
int g2048_sync(g2048_ctx* ctx);
int g2048_timer_start(g2048_ctx* ctx);                /* hipEventRecord on the context's stream */
int g2048_timer_stop(g2048_ctx* ctx, float* ms);      /* second event + elapsed time, synchronises */

/* ---- lane state I/O (Game.row / Game.score and the locals of QAgent.episode) */
int g2048_set_boards(g2048_ctx* ctx, const uint8_t* boards /* [B][16] */);
int g2048_get_boards(g2048_ctx* ctx, uint8_t* boards /* [B][16] */);
int g2048_set_scores(g2048_ctx* ctx, const int32_t* scores /* [B] */);
int g2048_get_scores(g2048_ctx* ctx, int32_t* scores /* [B] */);
int g2048_set_rng(g2048_ctx* ctx, const uint64_t* state /* [B][2] */);
int g2048_get_rng(g2048_ctx* ctx, uint64_t* state /* [B][2] */);
int g2048_get_carry(g2048_ctx* ctx, uint8_t* prev /* [B][16] */, float* label /* [B] */, uint8_t* flags /* [B] */);
int g2048_clear_carry(g2048_ctx* ctx);                /* state, old_label = None, 0 (r_learning.py:226); clears DONE */
int g2048_reset(g2048_ctx* ctx);                      /* new game in every lane from the lane's own stream */
int g2048_set_auto_reset(g2048_ctx* ctx, int on);     /* on (default): a finished lane starts a new game at once */

/* ---- environment */
/* Game.pre_move for all four directions (game_logic.py:136-142).  changed: bit d set iff direction d moves. */
int g2048_move_all(g2048_ctx* ctx, uint8_t* after /* [B][4][16] */, int32_t* reward /* [B][4] */, uint8_t* changed /* [B] */);
/* Game.make_move (game_logic.py:144-148) with a per-lane direction; moved[i] = 1 iff the board changed */
int g2048_apply_moves(g2048_ctx* ctx, const uint8_t* dirs /* [B] */, uint8_t* moved /* [B], may be NULL */);
/* Game.game_over / empty_count / adjacent_pair_count (game_logic.py:101-110); any output may be NULL */
int g2048_terminal(g2048_ctx* ctx, uint8_t* over, uint8_t* n_empty, uint8_t* n_pairs);
/* Game.new_tile (game_logic.py:112-121) from the lane streams; full boards are left alone (k = 255).
 * The (r10, k) draws used are exported so the same game can be replayed into the reference. */
int g2048_spawn(g2048_ctx* ctx, uint8_t* r10 /* [B] or NULL */, uint8_t* k /* [B] or NULL */);
int g2048_spawn_injected(g2048_ctx* ctx, const uint8_t* r10 /* [B] */, const uint8_t* k /* [B] */);
/* Stateless form of g2048_move_all for `count` caller-supplied boards (the lanes are not touched): Game.pre_move x 4
 * on arbitrary positions, e.g. the nodes of the look-ahead tree (game_logic.py:214-243).  changed == 0 with no empty
 * cell is game over. */
int g2048_boards_move_all(g2048_ctx* ctx, const uint8_t* boards /* [count][16] */, int64_t count, uint8_t* after /* [count][4][16] */,
                          int32_t* reward /* [count][4] */, uint8_t* changed /* [count] */);
/* nsteps x {uniformly random valid direction, move, spawn, terminal check / auto-reset}: BASELINE config 2 */
int g2048_step_random(g2048_ctx* ctx, uint32_t nsteps);

/* ---- features / value (need n_tuple != 0) */
int g2048_features(g2048_ctx* ctx, int32_t* out /* [B][F] */);            /* f_n, r_learning.py:17-69 (no offsets) */
int g2048_weights_set(g2048_ctx* ctx, const float* w, int64_t count);     /* flat, weight_signature group order */
int g2048_weights_get(g2048_ctx* ctx, float* w, int64_t count);
int g2048_weights_init(g2048_ctx* ctx, uint64_t seed, float scale);       /* U[0, scale): init_weights :139-149 uses 0.01 */
int g2048_evaluate(g2048_ctx* ctx, float* value /* [B] */);               /* QAgent.evaluate, r_learning.py:202-203 */
/* the same for `count` caller-supplied boards (the lanes are not touched) */
int g2048_boards_evaluate(g2048_ctx* ctx, const uint8_t* boards /* [count][16] */, int64_t count, float* value /* [count] */);
/* Game.look_forward (game_logic.py:214-243) — the reference's sampled expectimax — for `count` caller-supplied positions (the lanes
 * are not touched): value[i] = V_depth(boards[i]) with
 *     V_0(s) = evaluate(s);   V_d(s) = evaluate(s) if empty(s) >= since_empty, else the mean over min(width, empty(s)) sampled new
 *     tiles t of max(0, W(s + t));   W(c) = -100 if c is game over, else the max over the directions that change c of V_{d-1}.
 * All trees are expanded, evaluated and reduced level by level on the device (csrc/lookahead.hip).  The chance nodes of a position
 * are a function of its board and salt[i] (two words; NULL = zeros): the xoroshiro128++ stream of 2048_amd/rng.py
 * `lookahead_draws` — distinct empty cells without replacement, tile 2 with probability 1/10 — so the same draws can be fed to the
 * reference's recursion (tests/golden/make_golden4.py).  A full board (no sample possible) yields NaN where the reference divides
 * by zero.  depth 0 .. 6, width 1 .. 16, (4 width)^depth <= 2^22. */
int g2048_boards_look_forward(g2048_ctx* ctx, const uint8_t* boards /* [count][16] */, int64_t count, int depth, int width, int since_empty,
                              const uint64_t* salt /* [count][2] or NULL */, float* value /* [count] */);
/* nsteps x the body of Game.trial_run (game_logic.py:170-183) with look-ahead for every live lane: Game._find_best_move (:150-161:
 * first maximum of look_forward over the directions that change the board; the salt of a lane's chance nodes is its RNG state at
 * that move, no draw is consumed), Game._move_on (:163-167: the move, the new tile from the lane's stream), then the loop's tests:
 * a lane ends when its game is over or a tile >= 2^limit_tile is on the board (limit_tile 0: never) — counted in the statistics and
 * recorded in the game logs like a TD step's.  Nothing is learned.  depth 0 is the greedy choice.  ASYNCHRONOUS. */
int g2048_lookahead_steps(g2048_ctx* ctx, int depth, int width, int since_empty, int limit_tile, uint32_t nsteps);
/* greedy afterstate choice (r_learning.py:229-237, game_logic.py:150-161 at depth 0): first maximum over the
 * directions that change the board; action 255 / value 0 when none does.  values4 ([B][4], may be NULL) gets
 * V(afterstate d) or -inf.  value and action may both be NULL: the kernel runs and nothing is copied back. */
int g2048_eval_select(g2048_ctx* ctx, float* value /* [B] */, uint8_t* action /* [B] */, float* values4);

/* ---- learning */
/* QAgent.update (r_learning.py:207-214) for `count` (state, dw) records: += dw at every feature slot of the
 * 8 symmetric images (fp32 atomic adds). */
int g2048_update(g2048_ctx* ctx, const uint8_t* states /* [count][16] */, const float* dw /* [count] */, int64_t count);
/* nsteps synchronous board-steps of QAgent.episode (r_learning.py:228-249) for every live lane: all lanes
 * choose with the same table, then every (state, dw) record of the step is added.  alpha is used as given.
 * ASYNCHRONOUS: returns when the steps are queued (for batches >= 2^17 lanes the host stays at most one step ahead of
 * the device: it reads the update kernel's load statistics back after every step to keep its work plan balanced). */
int g2048_td_steps(g2048_ctx* ctx, float alpha, uint32_t nsteps);
/* Lane order in memory.  k_td_play is bound by the L1 misses of its table gathers, and lanes whose big tiles sit in the
 * same cells with the same values touch the same cache lines: with every > 0 each `every`-th TD step re-orders the lanes by a
 * 16-bit key — a hash of `value >> 1` of every tile above a threshold (32; n = 3: 128), by cell (G2048_SORT_VALUES=0: one bit
 * per cell) — with a hand-written counting sort, three launches; the step that applies it reads its lanes through the
 * permutation.  Invisible through this ABI — every entry point that addresses lanes by index restores the identity order first
 * (one copy pass) — and to the results: a lane's game does not depend on where it sits.  Default 8 (G2048_SORT_EVERY); 0: never.
 * Batches below 2^17 lanes (G2048_SORT_MIN) and n = 2 keep their order. */
int g2048_set_lane_sort(g2048_ctx* ctx, uint32_t every);
/* test hook: the permutation the re-order would apply to the current boards (position i takes the lane at perm[i]) and the
 * key of every position; the lane order is left alone */
int g2048_debug_lane_order(g2048_ctx* ctx, uint32_t* perm /* [B] */, uint16_t* keys /* [B] */);
/* how the step's records are added to the table: 1 (default) = LDS-owner kernel (workgroups own 128 KiB table
 * slices in LDS, no global atomics for n <= 5), 0 = one global fp32 atomic per slot.  Same sums either way. */
int g2048_set_update_mode(g2048_ctx* ctx, int mode);
/* what a step's records do to a slot: 0 (default) = every dw is added, as QAgent.update does (r_learning.py:207-214);
 * 1 = per-slot mean: a slot that S of the step's dw target, C in number, moves by S / C.  Rule 1 is an extension for
 * large synchronous batches (with rule 0 the step size must shrink with the batch); it is not the
 * reference's arithmetic even at batch 1 (a symmetric board hits one slot several times). */
int g2048_set_update_rule(g2048_ctx* ctx, int rule);
/* the same, with HIP events around each of the step's two kernels (synchronises every step): average
 * milliseconds per launch of k_td_play and k_td_update, for the roofline line of bench.py */
int g2048_td_steps_profiled(g2048_ctx* ctx, float alpha, uint32_t nsteps, float* ms_play, float* ms_update);
/* per kernel: out4 = average ms per launch of {k_td_play, k_td_update_owner (both passes under rule 1), k_td_update_tail
 * (n = 6), k_apply_*} */
int g2048_td_steps_kernel_ms(g2048_ctx* ctx, float alpha, uint32_t nsteps, float* out4);

/* Diagnostics of the LDS-owner update (update mode 1): for each workgroup of the current plan six words
 * {orbit variant, chunk, part, nparts, start clock, end clock} of its last launch (clock: 100 MHz constant counter).
 * No reference counterpart. */
int g2048_debug_owner_plan(g2048_ctx* ctx, uint64_t* out, uint32_t capacity, uint32_t* count);
/* what every lane did in the latest TD step — the entries Game.moves / Game.tiles get in QAgent.episode
 * (r_learning.py:244, game_logic.py:121): bits 0-1 direction, bit 2 a move was made, bits 4-7 cell (4*r + c) of
 * the new tile, bits 8-9 the new tile (1 or 2), bit 10 a tile was placed, bit 11 the game ended on this step. */
int g2048_get_last_move(g2048_ctx* ctx, uint16_t* out /* [B] */);
/* Per-lane game records for the first `lanes` lanes (Game.moves / Game.tiles / starting_position, game_logic.py:55-66):
 * every TD step appends the lane's g2048_get_last_move word; two slots per lane, so the game that just ended stays
 * readable while the next one is written.  capacity = moves kept per game.  (0, 0) turns it off.
 * meta: [lanes][8] = slot in use, moves so far, games finished, length0, score0, length1, score1, flags
 * (bit s: slot s did not start at move 0; bit 2+s: slot s has more moves than `capacity`). */
int g2048_log_enable(g2048_ctx* ctx, uint32_t lanes, uint32_t capacity);
int g2048_log_meta(g2048_ctx* ctx, uint32_t* meta /* [lanes][8] */);
int g2048_log_game(g2048_ctx* ctx, uint32_t lane, uint32_t slot, uint16_t* moves /* [capacity] */, uint8_t* start /* [16] */);
/* the board a finished recorded game ended on (Game.row of the game QAgent.episode returns, r_learning.py:252) */
int g2048_log_final(g2048_ctx* ctx, uint32_t lane, uint32_t slot, uint8_t* board /* [16] */);
int g2048_stats_get(g2048_ctx* ctx, g2048_stats* out);
int g2048_stats_reset(g2048_ctx* ctx);

/* ---- multi-GPU (the reference trains in one Python thread, r_learning.py:269-296, application.py:611).
 * One process and one context per GPU.  Episodes are sharded by lane0 (no data-path collective); the table is replicated.
 * An epoch is E board-steps; g2048_delta_begin starts the first one (W0 = W); the epoch's delta is D = W - W0, formed when
 * it is asked for (G2048_DELTA_ACCUM=1 in the environment: an fp32 accumulator that mirrors every add instead).  At the end
 * of an epoch the ranks exchange D with ONE sum all-reduce over xGMI and every replica becomes
 *     rule 0 (sum):   W = W0 + sum_r D_r
 *     rule 1 (mean):  W = W0 + sum_r D_r / max(1, #{r : D_r != 0})   per slot — the mean over the ranks that moved the slot
 *                     (the payload is [D | touched], twice the table)
 * and W0 = W, D = 0 for the next epoch.
 *   native path:  g2048_comm_unique_id on rank 0 -> the 128 bytes reach the other ranks out of band -> g2048_comm_init on
 *                 every rank (collective) -> loop { g2048_td_steps(E); g2048_allreduce_deltas } — RCCL (ncclAllReduce) on the
 *                 context's stream, no host wait.  librccl is bound at run time (dlopen): G2048_ERR_COMM if it is absent.
 *   host-driven:  delta_extract -> any all-reduce of the caller (torch.distributed) -> delta_apply; these three synchronise
 *                 the context's stream.  dst/src are DEVICE pointers to fp32[table_slots] owned by the caller, or NULL for
 *                 the context's own accumulator (g2048_delta_device_ptr).  Mean rule: delta_pack_touched(pack fp32[2 * slots])
 *                 -> all-reduce(pack) -> delta_apply_mean(pack).
 * Order of entries: every buffer the CALLER owns (g2048_weights_get / _set, dst / src / pack above) is in the reference's
 * index order (g2048_feature_layout).  The table and the accumulator themselves are stored in another order (n >= 4: see
 * table_place / hex_place in csrc/features.hpp): the raw pointers g2048_weights_device_ptr and
 * g2048_delta_device_ptr are good for element-wise work between replicas (broadcast, all-reduce) only. */
#define G2048_COMM_ID_BYTES 128
int g2048_comm_unique_id(uint8_t* id /* [G2048_COMM_ID_BYTES] */);
int g2048_comm_init(g2048_ctx* ctx, int rank, int nranks, const uint8_t* id /* [G2048_COMM_ID_BYTES] */);
int g2048_comm_destroy(g2048_ctx* ctx);
/* what the communicator itself reports (ncclCommUserRank / ncclCommCount) — rank 0 of 1 without a communicator; lets the
 * record of a multi-GPU run prove how many ranks RCCL saw */
int g2048_comm_info(g2048_ctx* ctx, int* rank, int* nranks);
int g2048_allreduce_deltas(g2048_ctx* ctx);
/* sum (op_max = 0) or maximum (1) of `count` host doubles over the ranks, in place: episode statistics, timings */
int g2048_allreduce_f64(g2048_ctx* ctx, double* values, int count, int op_max);
int g2048_weights_device_ptr(g2048_ctx* ctx, void** ptr, int64_t* count);
int g2048_delta_begin(g2048_ctx* ctx);
int g2048_delta_extract(g2048_ctx* ctx, void* dst);
int g2048_delta_apply(g2048_ctx* ctx, const void* src);
int g2048_delta_pack_touched(g2048_ctx* ctx, void* pack /* device fp32[2 * table_slots] */);
int g2048_delta_apply_mean(g2048_ctx* ctx, const void* pack);
int g2048_delta_device_ptr(g2048_ctx* ctx, void** ptr);
int g2048_stream_handle(g2048_ctx* ctx, void** hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* G2048_H */
