// cpu_ref.cpp — lib2048_cpu.so: the SAME C ABI (include/g2048.h) on the host, scalar C++ (last
// paragraph).
//
// What it is for: the Python surface (Game / QAgent, show.py's calls) on a box without a GPU, a parity target that runs
// under `pytest -m "not gpu"`, and the second, stronger CPU line of bench.py (1 thread and all threads) beside the
// NumPy-structured port of the reference.  What it is NOT: a fallback.  It is selected explicitly (G2048_BACKEND=cpu or
// Engine(backend='cpu')); the HIP library never routes here, and nothing here is the oracle (oracle/ is test
// infrastructure: this file shares no code with it).
//
// It is built from the build's own integer logic — board_ops.hpp (SWAR slide / merge / score, spawn, terminal test,
// xoroshiro128++ lanes) and features.hpp (f_2 .. f_6 index encoders, D4 images) — the very headers the HIP kernels
// include, compiled for the host.  The floating-point statements are the device's, operation for operation (value =
// left-to-right fp32 sum; dw = ((float)reward + V - old_label) * alpha / F; -ffp-contract=off), so moves, scores, RNG
// streams, records and greedy choices are bit-identical to the GPU's; a step's table sums are formed in float64 per slot
// and added to the fp32 table once (the GPU sums in 64-bit fixed point and adds once: both are within the parity tests'
// fp32-accumulation tolerance of the float64 oracle).
//
// The table is stored in the reference's index order (no table_place / hex_place: those are cache-line layouts of the GPU).
// Threads: OpenMP over lanes (step part 1) and over records (step part 2, atomic float64 adds); G2048_CPU_THREADS, default 1.
// Multi-GPU group: the epoch-delta entry points work on HOST pointers (so the two-rank gloo job runs end to end on this
// backend); g2048_comm_* / g2048_allreduce_* answer G2048_ERR_COMM (there is no RCCL here).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <memory>
#include <string>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

#include "../../include/g2048.h"
#include "features.hpp"

using namespace g2048;

namespace {

constexpr uint8_t HAS_PREV = G2048_LANE_HAS_PREV, DONE = G2048_LANE_DONE;

struct Table {      // the weight table and the per-step accumulators that go with it (shared by g2048_create_shared contexts)
    std::vector<float> w;
    std::vector<double> acc;        // this step's sum of dw per slot
    std::vector<uint32_t> cnt;      // this step's number of adds per slot
    std::vector<uint32_t> touched;  // slots with cnt > 0
    std::vector<float> w0;          // multi-GPU epoch: the table at the start of the epoch
    std::vector<float> delta;       // the context's own delta buffer (g2048_delta_extract(NULL), g2048_delta_device_ptr)
    bool tracking = false;
};

struct Record {
    Board state;
    float dw;
};

struct GameLogHost {
    uint32_t lanes = 0, capacity = 0;
    std::vector<uint16_t> moves;
    std::vector<Board> start, final;
    std::vector<uint32_t> meta;
};

}  // namespace

struct g2048_ctx {
    int n = 0, F = 0, auto_reset = 1, update_rule = 0, update_mode = 1, threads = 1;
    uint32_t B = 0;
    uint64_t seed = 0, lane0 = 0, slots = 0;
    g2048_ctx* parent = nullptr;
    std::shared_ptr<Table> table;
    std::vector<Board> boards, prev;
    std::vector<int32_t> scores;
    std::vector<Rng> rng;
    std::vector<float> label;
    std::vector<uint8_t> flags;
    std::vector<uint16_t> last_move;
    std::vector<Record> recs;       // scratch of a TD step
    g2048_stats stats{};
    GameLogHost log;
    std::chrono::steady_clock::time_point t0;
    std::string err;
};

namespace {

int fail(g2048_ctx* c, int code, const char* what) {
    if (c) c->err = what;
    return code;
}
#define NEED(c, cond, msg) \
    do {                   \
        if (!(cond)) return fail(c, G2048_ERR_ARG, msg); \
    } while (0)
#define NEED_TABLE(c) \
    do {              \
        if ((c)->n == 0) return fail(c, G2048_ERR_STATE, "this context has no weight table (n_tuple == 0)"); \
    } while (0)

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

#define BY_N(c, EXPR)                                \
    switch ((c)->n) {                                \
        case 2: { constexpr int N = 2; EXPR; } break; \
        case 3: { constexpr int N = 3; EXPR; } break; \
        case 4: { constexpr int N = 4; EXPR; } break; \
        case 5: { constexpr int N = 5; EXPR; } break; \
        case 6: { constexpr int N = 6; EXPR; } break; \
        default: break;                              \
    }

Board load_board(const uint8_t* p) {
    Board b;
    memcpy(b.r, p, 16);     // byte j of word r = cell (r, j): little-endian row words ARE the 16 row-major bytes
    return b;
}
void store_board(uint8_t* p, const Board& b) { memcpy(p, b.r, 16); }

struct Moves4 {
    Moved m[4];
};
Moves4 all_moves(const Board& b) {
    uint32_t cols[4];
    transpose(b.r, cols);
    Moves4 r;
    r.m[0] = move_dir<0>(b.r, cols);
    r.m[1] = move_dir<1>(b.r, cols);
    r.m[2] = move_dir<2>(b.r, cols);
    r.m[3] = move_dir<3>(b.r, cols);
    return r;
}
uint32_t changed_mask(const Moves4& mv) {
    return (mv.m[0].changed ? 1u : 0u) | (mv.m[1].changed ? 2u : 0u) | (mv.m[2].changed ? 4u : 0u) | (mv.m[3].changed ? 8u : 0u);
}

// QAgent.evaluate (r_learning.py:202-203): a left-to-right fp32 sum of one weight per feature
template <int N>
float value_of(const float* w, const Board& b) {
    constexpr int F = Shape<N>::F;
    uint32_t s[F];
    feature_slots<N>(pack_board(b), s);
    float v = 0.0f;
    for (int f = 0; f < F; ++f) v += w[s[f]];
    return v;
}

struct Choice {
    int action;
    float value;
    float v[4];
};
// greedy afterstate choice (r_learning.py:229-237): strict '>' from -inf keeps the first maximum
template <int N>
Choice choose(const float* w, const Moves4& mv) {
    Choice c;
    c.action = -1;
    c.value = -INFINITY;
    int first_valid = -1;
    for (int d = 0; d < 4; ++d) {
        c.v[d] = -INFINITY;
        if (!mv.m[d].changed) continue;
        if (first_valid < 0) first_valid = d;
        c.v[d] = value_of<N>(w, mv.m[d].after);
        if (c.v[d] > c.value) {
            c.value = c.v[d];
            c.action = d;
        }
    }
    if (c.action < 0 && first_valid >= 0) {     // every value was -inf or NaN (a poisoned table): still make a legal move
        c.action = first_valid;
        c.value = c.v[first_valid];
    }
    return c;
}

void count_finished(g2048_stats& st, const Board& b, int32_t score, bool overflow) {
    const uint64_t sc = score < 0 ? 0u : (uint64_t)score;
    st.episodes += 1;
    st.score_sum += sc;
    st.best_score = std::max(st.best_score, sc);
    const uint32_t t = max_tile(b);
    st.max_tile[t > 19u ? 19u : t] += 1;
    if (overflow) st.overflow16 += 1;
}
void merge_stats(g2048_stats& into, const g2048_stats& s) {
    into.episodes += s.episodes;
    into.moves += s.moves;
    into.score_sum += s.score_sum;
    into.best_score = std::max(into.best_score, s.best_score);
    for (int i = 0; i < 20; ++i) into.max_tile[i] += s.max_tile[i];
    into.overflow16 += s.overflow16;
    into.nonfinite += s.nonfinite;
    into.valid_dirs += s.valid_dirs;
}

constexpr uint32_t LOG_PARTIAL0 = 1u, LOG_TRUNC0 = 4u;
void log_init(g2048_ctx* c) {
    GameLogHost& lg = c->log;
    for (uint32_t i = 0; i < lg.lanes; ++i) {
        uint32_t* m = lg.meta.data() + 8 * (size_t)i;
        for (int j = 0; j < 8; ++j) m[j] = 0;
        m[7] = (c->flags[i] & HAS_PREV) ? LOG_PARTIAL0 : 0u;
        lg.start[2 * (size_t)i] = c->boards[i];
    }
}
void log_step(GameLogHost& lg, uint32_t i, uint32_t lm, bool moved, bool over, int32_t final_score, bool restarted, const Board& fresh,
              const Board& last) {
    uint32_t* m = lg.meta.data() + 8 * (size_t)i;
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
        lg.final[(size_t)i * 2 + slot] = last;
        if (restarted) {
            slot ^= 1u;
            cnt = 0;
            m[3 + 2 * slot] = 0;
            m[7] &= ~((LOG_PARTIAL0 | LOG_TRUNC0) << slot);
            lg.start[(size_t)i * 2 + slot] = fresh;
        }
    }
    m[0] = slot;
    m[1] = cnt;
}

// ---- a step's records -> the table.  rule 0: every dw is added (QAgent.update, r_learning.py:207-214: all 8 images);
// rule 1: a slot moves by the mean of the dw that target it.
// Threads: every thread sums its share of the records in a private direct-mapped cache first ({slot, count, float64 sum},
// 2^16 entries = 1 MB, stays in the core's L2) and only evictions and the final flush touch the shared arrays with atomics:
// real boards put a large share of a step's adds on a few slots (an empty line is index 0), and 64 threads hammering those
// with atomic adds ran SLOWER than one thread (92 k against 541 k board-steps/s, round-3 bench host).
struct CacheLine {
    uint32_t slot, cnt;
    double sum;
};
constexpr uint32_t CACHE_BITS = 16, CACHE_EMPTY = 0xFFFFFFFFu;

inline void flush_line(Table& t, CacheLine& e, std::vector<uint32_t>& fresh) {
    uint32_t before;
#pragma omp atomic capture
    {
        before = t.cnt[e.slot];
        t.cnt[e.slot] += e.cnt;
    }
#pragma omp atomic update
    t.acc[e.slot] += e.sum;
    if (before == 0) fresh.push_back(e.slot);
    e.slot = CACHE_EMPTY;
}

template <int N>
void accumulate(g2048_ctx* c, const Record* recs, size_t count) {
    constexpr int F = Shape<N>::F;
    Table& t = *c->table;
    if (t.acc.empty()) {
        t.acc.assign(c->slots, 0.0);
        t.cnt.assign(c->slots, 0u);
    }
    const int nthreads = c->threads;
    std::vector<std::vector<uint32_t>> fresh((size_t)nthreads);
    if (nthreads == 1) {        // one thread: straight into the shared arrays
        std::vector<uint32_t>& mine = fresh[0];
        for (size_t r = 0; r < count; ++r) {
            const Packed p = pack_board(recs[r].state);
            const double dw = (double)recs[r].dw;
            for (uint32_t g = 0; g < 8; ++g) {
                uint32_t s[F];
                feature_slots<N>(d4_image(p, g), s);
                for (int f = 0; f < F; ++f) {
                    if (t.cnt[s[f]]++ == 0) mine.push_back(s[f]);
                    t.acc[s[f]] += dw;
                }
            }
        }
    } else {
#pragma omp parallel num_threads(nthreads)
        {
#ifdef _OPENMP
            const size_t tid = (size_t)omp_get_thread_num();
#else
            const size_t tid = 0;
#endif
            std::vector<uint32_t>& mine = fresh[tid];
            std::vector<CacheLine> cache((size_t)1 << CACHE_BITS, CacheLine{CACHE_EMPTY, 0u, 0.0});
#pragma omp for schedule(static)
            for (long long r = 0; r < (long long)count; ++r) {
                const Packed p = pack_board(recs[r].state);
                const double dw = (double)recs[r].dw;
                for (uint32_t g = 0; g < 8; ++g) {
                    uint32_t s[F];
                    feature_slots<N>(d4_image(p, g), s);
                    for (int f = 0; f < F; ++f) {
                        CacheLine& e = cache[(s[f] * 2654435761u) >> (32 - CACHE_BITS)];
                        if (e.slot != s[f]) {
                            if (e.slot != CACHE_EMPTY) flush_line(t, e, mine);
                            e = CacheLine{s[f], 0u, 0.0};
                        }
                        e.cnt += 1;
                        e.sum += dw;
                    }
                }
            }
            for (CacheLine& e : cache)
                if (e.slot != CACHE_EMPTY) flush_line(t, e, mine);
        }
    }
    for (auto& v : fresh) t.touched.insert(t.touched.end(), v.begin(), v.end());
}

void apply_accumulated(g2048_ctx* c) {
    Table& t = *c->table;
    const int rule = c->update_rule;
    const size_t n = t.touched.size();
#pragma omp parallel for num_threads(c->threads) schedule(static)
    for (long long j = 0; j < (long long)n; ++j) {
        const uint32_t s = t.touched[(size_t)j];
        const double sum = t.acc[s];
        t.w[s] += rule == 1 ? (float)(sum / (double)t.cnt[s]) : (float)sum;
        t.acc[s] = 0.0;
        t.cnt[s] = 0u;
    }
    t.touched.clear();
}

// One synchronous board-step of every live lane: the body of `while not game.game_over` in QAgent.episode
// (r_learning.py:228-249), statement for statement what k_td_play does per lane (g2048.hip, td_play_body).
template <int N>
void td_step(g2048_ctx* c, float alpha) {
    constexpr float F = (float)Shape<N>::F;
    const uint32_t B = c->B;
    const float* w = c->table->w.data();
    c->recs.resize(B);
    const int nthreads = c->threads;
    std::vector<std::vector<Record>> terminal((size_t)nthreads);
    std::vector<g2048_stats> part((size_t)nthreads);
    for (auto& s : part) memset(&s, 0, sizeof s);
    GameLogHost& lg = c->log;
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (long long ii = 0; ii < (long long)B; ++ii) {
        const uint32_t i = (uint32_t)ii;
#ifdef _OPENMP
        const size_t tid = (size_t)omp_get_thread_num();
#else
        const size_t tid = 0;
#endif
        g2048_stats& st = part[tid];
        uint8_t fl = c->flags[i];
        float dw1 = 0.0f;
        uint32_t lm = 0;
        c->recs[i].state = c->prev[i];      // the record's state is the one chosen in the PREVIOUS step
        if (!(fl & DONE)) {
            Board b = c->boards[i];
            Rng g = c->rng[i];
            int32_t score = c->scores[i];
            float old_label = c->label[i];
            const Moves4 mv = all_moves(b);
            const Choice ch = choose<N>(w, mv);
            bool over, overflow = false, moved = false;
            if (ch.action >= 0) {
                const Moved& m = mv.m[ch.action];
                const int32_t reward = (int32_t)merged_score(m.ma, m.mb);
                if (fl & HAS_PREV) dw1 = ((float)reward + ch.value - old_label) * alpha / F;
                score += reward;
                c->prev[i] = m.after;
                old_label = ch.value;
                fl |= HAS_PREV;
                moved = true;
                st.moves += 1;
                st.valid_dirs += popcount32(changed_mask(mv));
                b = m.after;
                lm = (uint32_t)ch.action | 4u;
                if (spawn(b, g)) {
                    for (int r = 0; r < 4; ++r) {
                        const uint32_t d = b.r[r] ^ m.after.r[r];       // the one byte that changed
                        if (d) {
                            const uint32_t col = (uint32_t)__builtin_ctz(d) >> 3;
                            lm |= ((uint32_t)(4 * r) + col) << 4 | ((d >> (8 * col)) & 3u) << 8 | 1u << 10;
                        }
                    }
                }
                overflow = max_tile(b) >= 16u;
                over = game_over(b) || overflow;
                if (over) {
                    const float dw2 = -ch.value * alpha / F;
                    if (std::isfinite(dw2))
                        terminal[tid].push_back(Record{m.after, dw2});
                    else
                        st.nonfinite += 1;
                }
            } else {            // a dead board was loaded: only the terminal update remains
                over = true;
                if (fl & HAS_PREV) {
                    const float dw2 = -old_label * alpha / F;
                    if (std::isfinite(dw2))
                        terminal[tid].push_back(Record{c->prev[i], dw2});
                    else
                        st.nonfinite += 1;
                }
            }
            const int32_t final_score = score;
            const Board final_board = b;
            if (over) {
                lm |= 1u << 11;
                count_finished(st, b, score, overflow);
                if (c->auto_reset) {
                    b = new_game(g);
                    score = 0;
                    old_label = 0.0f;
                    fl &= (uint8_t)~HAS_PREV;
                } else {
                    fl |= DONE;
                }
            }
            if (i < lg.lanes) log_step(lg, i, lm, moved, over, final_score, over && c->auto_reset, b, final_board);
            c->boards[i] = b;
            c->rng[i] = g;
            c->scores[i] = score;
            c->label[i] = old_label;
            c->flags[i] = fl;
        }
        if (!std::isfinite(dw1)) {
            st.nonfinite += 1;
            dw1 = 0.0f;
        }
        c->recs[i].dw = dw1;
        c->last_move[i] = (uint16_t)lm;
    }
    for (const auto& s : part) merge_stats(c->stats, s);
    // step part 2: the records (dw = 0: none) and the terminal queue
    size_t kept = 0;
    for (uint32_t i = 0; i < B; ++i)
        if (c->recs[i].dw != 0.0f) c->recs[kept++] = c->recs[i];
    c->recs.resize(kept);
    for (auto& q : terminal) c->recs.insert(c->recs.end(), q.begin(), q.end());
    if (c->update_mode == 0) {      // one fp32 add per slot and image, as the global-atomics kernel (k_td_update) makes them
        float* wt = c->table->w.data();
        for (const Record& r : c->recs) {
            const Packed p = pack_board(r.state);
            for (uint32_t g = 0; g < 8; ++g) {
                uint32_t s[Shape<N>::F];
                feature_slots<N>(d4_image(p, g), s);
                for (int f = 0; f < Shape<N>::F; ++f) wt[s[f]] += r.dw;
            }
        }
        return;
    }
    accumulate<N>(c, c->recs.data(), c->recs.size());
    apply_accumulated(c);
}

template <int N>
void features_impl(g2048_ctx* c, int32_t* out) {
    constexpr int F = Shape<N>::F;
    for (uint32_t i = 0; i < c->B; ++i) {
        uint32_t s[F];
        feature_slots<N>(pack_board(c->boards[i]), s);
        for (int f = 0; f < F; ++f) out[(size_t)i * F + f] = (int32_t)(s[f] - feature_offset(N, f));
    }
}

template <int N>
void evaluate_impl(g2048_ctx* c, const Board* boards, const uint8_t* bytes, int64_t count, float* value) {
    const float* w = c->table->w.data();
#pragma omp parallel for num_threads(c->threads) schedule(static)
    for (long long i = 0; i < (long long)count; ++i) value[i] = value_of<N>(w, boards ? boards[i] : load_board(bytes + 16 * i));
}

template <int N>
void eval_select_impl(g2048_ctx* c, float* value, uint8_t* action, float* values4) {
    const float* w = c->table->w.data();
#pragma omp parallel for num_threads(c->threads) schedule(static)
    for (long long i = 0; i < (long long)c->B; ++i) {
        const Choice ch = choose<N>(w, all_moves(c->boards[(size_t)i]));
        value[i] = ch.action < 0 ? 0.0f : ch.value;
        action[i] = ch.action < 0 ? (uint8_t)255 : (uint8_t)ch.action;
        if (values4)
            for (int d = 0; d < 4; ++d) values4[4 * i + d] = ch.v[d];
    }
}

// QAgent.update (r_learning.py:207-214): fp32 adds at every feature slot of the 8 images, in record order
template <int N>
void update_impl(g2048_ctx* c, const uint8_t* states, const float* dw, int64_t count) {
    constexpr int F = Shape<N>::F;
    float* w = c->table->w.data();
    for (int64_t r = 0; r < count; ++r) {
        const Packed p = pack_board(load_board(states + 16 * r));
        for (uint32_t g = 0; g < 8; ++g) {
            uint32_t s[F];
            feature_slots<N>(d4_image(p, g), s);
            for (int f = 0; f < F; ++f) w[s[f]] += dw[r];
        }
    }
}

}  // namespace

// ---- look-ahead: Game.look_forward (game_logic.py:214-243) as the scalar recursion it is, with the chance nodes of the
// device's spec (2048_amd/rng.py: lookahead_draws — a function of the node's board and a salt) and its fp32 arithmetic
namespace {
struct Salt {
    uint64_t s0, s1;
};
Rng la_rng(const Board& b, const Salt& salt) {
    const uint64_t lo = (uint64_t)b.r[0] | (uint64_t)b.r[1] << 32, hi = (uint64_t)b.r[2] | (uint64_t)b.r[3] << 32;
    uint64_t x = salt.s0 ^ lo;
    const uint64_t a = splitmix64(x);
    x = (x ^ hi) + salt.s1;
    Rng g;
    g.s0 = splitmix64(x);
    x ^= a;
    g.s1 = splitmix64(x);
    if ((g.s0 | g.s1) == 0) g.s0 = 1;
    return g;
}
template <int N>
float look_forward(const float* w, const Board& b, int depth, int width, int since_empty, const Salt& salt) {
    if (depth == 0) return value_of<N>(w, b);
    uint32_t free_cells = empty_bits(b);
    const uint32_t ne = popcount32(free_cells);
    if ((int)ne >= since_empty) return value_of<N>(w, b);
    const uint32_t k = ne < (uint32_t)width ? ne : (uint32_t)width;
    if (k == 0) return NAN;                     // a full board: the reference divides by zero (:242)
    Rng g = la_rng(b, salt);
    float acc = 0.0f;
    for (uint32_t j = 0; j < k; ++j) {
        uint32_t r10, kk;
        spawn_draw(next_u64(g), popcount32(free_cells), r10, kk);
        Board child = b;
        place_tile(child, r10, kk, free_cells);
        free_cells &= ~(1u << kth_set_bit(free_cells, kk));
        const Moves4 mv = all_moves(child);
        float best = -INFINITY;
        bool any = false;
        for (int d = 0; d < 4; ++d)
            if (mv.m[d].changed) {
                any = true;
                const float v = look_forward<N>(w, mv.m[d].after, depth - 1, width, since_empty, salt);
                if (v > best) best = v;
            }
        const float wj = any ? best : -100.0f;
        acc += (0.0f > wj) ? 0.0f : wj;
    }
    return acc / (float)k;
}
int la_args(g2048_ctx* c, int depth, int width, int since_empty, int limit_tile) {
    NEED(c, depth >= 0 && depth <= 6, "look-ahead depth out of range (0 .. 6)");
    NEED(c, width >= 1 && width <= 16, "look-ahead width out of range (1 .. 16)");
    NEED(c, since_empty >= 0 && limit_tile >= 0, "negative since_empty / limit_tile");
    double leaves = 1;
    for (int l = 0; l < depth; ++l) leaves *= 4.0 * width;
    NEED(c, leaves <= 4194304.0, "look-ahead tree too large: (4 width)^depth must stay below 2^22 nodes per position");
    return G2048_OK;
}
}  // namespace

namespace {
template <int N>
void lookahead_step(g2048_ctx* c, int depth, int width, int since_empty, uint32_t limit_tile) {
    const float* w = c->table->w.data();
    const int nthreads = c->threads;
    (void)nthreads;
    std::vector<g2048_stats> part((size_t)std::max(1, nthreads));
    GameLogHost& lg = c->log;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
    for (long long ii = 0; ii < (long long)c->B; ++ii) {
        const uint32_t i = (uint32_t)ii;
#ifdef _OPENMP
        g2048_stats& st = part[(size_t)omp_get_thread_num()];
#else
        g2048_stats& st = part[0];
#endif
        uint8_t fl = c->flags[i];
        if (fl & DONE) {
            c->last_move[i] = 0;
            continue;
        }
        Board b = c->boards[i];
        Rng g = c->rng[i];
        int32_t score = c->scores[i];
        const Salt salt{g.s0, g.s1};
        int action = -1, first_valid = -1;
        float best = -INFINITY;
        const Moves4 mv = all_moves(b);
        if (!(limit_tile && max_tile(b) >= limit_tile))
            for (int d = 0; d < 4; ++d)
                if (mv.m[d].changed) {
                    if (first_valid < 0) first_valid = d;
                    const float v = look_forward<N>(w, mv.m[d].after, depth, width, since_empty, salt);
                    if (v > best) {
                        best = v;
                        action = d;
                    }
                }
        if (action < 0) action = first_valid;
        uint32_t lm = 0;
        bool moved = false, over, overflow = false;
        if (action >= 0) {
            const Moved& m = mv.m[action];
            score += (int32_t)merged_score(m.ma, m.mb);
            b = m.after;
            moved = true;
            st.moves += 1;
            st.valid_dirs += popcount32(changed_mask(mv));
            lm = (uint32_t)action | 4u;
            if (spawn(b, g)) {
                for (int r = 0; r < 4; ++r) {
                    const uint32_t d = b.r[r] ^ m.after.r[r];
                    if (d) {
                        const uint32_t col = (uint32_t)__builtin_ctz(d) >> 3;
                        lm |= ((uint32_t)(4 * r) + col) << 4 | ((d >> (8 * col)) & 3u) << 8 | 1u << 10;
                    }
                }
            }
            overflow = max_tile(b) >= 16u;
            over = game_over(b) || overflow || (limit_tile && max_tile(b) >= limit_tile);
        } else {
            over = true;
        }
        const int32_t final_score = score;
        const Board final_board = b;
        if (over) {
            lm |= 1u << 11;
            count_finished(st, b, score, overflow);
            if (c->auto_reset) {
                b = new_game(g);
                score = 0;
            } else {
                fl |= DONE;
            }
        }
        if (i < lg.lanes) log_step(lg, i, lm, moved, over, final_score, over && c->auto_reset, b, final_board);
        c->boards[i] = b;
        c->rng[i] = g;
        c->scores[i] = score;
        c->flags[i] = fl;
        c->last_move[i] = (uint16_t)lm;
    }
    for (const auto& s : part) merge_stats(c->stats, s);
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
    *count = 1;         // "device 0" of this backend is the host
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

static int create_impl(uint32_t batch, int n, uint64_t seed, uint64_t lane0, g2048_ctx* parent, g2048_ctx** out) {
    int F;
    uint64_t slots;
    if (!out || batch == 0 || shape_of(n, &F, &slots) != 0) return G2048_ERR_ARG;
    g2048_ctx* c = new (std::nothrow) g2048_ctx;
    if (!c) return G2048_ERR_NOMEM;
    c->n = n;
    c->F = F;
    c->slots = slots;
    c->B = batch;
    c->seed = seed;
    c->lane0 = lane0;
    c->parent = parent;
    if (const char* e = getenv("G2048_CPU_THREADS")) c->threads = std::max(1, atoi(e));
#ifndef _OPENMP
    c->threads = 1;
#endif
    try {
        if (parent) {
            c->table = parent->table;
        } else if (n) {
            c->table = std::make_shared<Table>();
            c->table->w.assign(slots, 0.0f);
        }
        c->boards.resize(batch);
        c->prev.assign(batch, Board{{0, 0, 0, 0}});
        c->scores.assign(batch, 0);
        c->rng.resize(batch);
        c->label.assign(batch, 0.0f);
        c->flags.assign(batch, 0);
        c->last_move.assign(batch, 0);
    } catch (const std::bad_alloc&) {
        delete c;
        return G2048_ERR_NOMEM;
    }
    for (uint32_t i = 0; i < batch; ++i) {
        c->rng[i] = seed_lane(seed, lane0 + i);
        c->boards[i] = new_game(c->rng[i]);
    }
    *out = c;
    return G2048_OK;
}

int g2048_create(int device, uint32_t batch, int n_tuple, uint64_t seed, uint64_t lane0, g2048_ctx** out) {
    if (device != 0) return G2048_ERR_ARG;          // the host is device 0 of this backend
    return create_impl(batch, n_tuple, seed, lane0, nullptr, out);
}

int g2048_create_shared(g2048_ctx* parent, uint32_t batch, uint64_t seed, uint64_t lane0, g2048_ctx** out) {
    if (!parent || parent->n == 0) return G2048_ERR_ARG;
    return create_impl(batch, parent->n, seed, lane0, parent, out);
}

int g2048_destroy(g2048_ctx* c) {
    delete c;
    return G2048_OK;
}

int g2048_sync(g2048_ctx* c) { return c ? G2048_OK : G2048_ERR_ARG; }

int g2048_timer_start(g2048_ctx* c) {
    if (!c) return G2048_ERR_ARG;
    c->t0 = std::chrono::steady_clock::now();
    return G2048_OK;
}

int g2048_timer_stop(g2048_ctx* c, float* ms) {
    if (!c || !ms) return G2048_ERR_ARG;
    *ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - c->t0).count();
    return G2048_OK;
}

int g2048_set_boards(g2048_ctx* c, const uint8_t* src) {
    if (!c || !src) return c ? fail(c, G2048_ERR_ARG, "null buffer") : G2048_ERR_ARG;
    for (uint32_t i = 0; i < c->B; ++i) c->boards[i] = load_board(src + 16 * (size_t)i);
    return G2048_OK;
}
int g2048_get_boards(g2048_ctx* c, uint8_t* dst) {
    if (!c || !dst) return c ? fail(c, G2048_ERR_ARG, "null buffer") : G2048_ERR_ARG;
    for (uint32_t i = 0; i < c->B; ++i) store_board(dst + 16 * (size_t)i, c->boards[i]);
    return G2048_OK;
}
int g2048_set_scores(g2048_ctx* c, const int32_t* src) {
    if (!c || !src) return c ? fail(c, G2048_ERR_ARG, "null buffer") : G2048_ERR_ARG;
    memcpy(c->scores.data(), src, (size_t)c->B * 4);
    return G2048_OK;
}
int g2048_get_scores(g2048_ctx* c, int32_t* dst) {
    if (!c || !dst) return c ? fail(c, G2048_ERR_ARG, "null buffer") : G2048_ERR_ARG;
    memcpy(dst, c->scores.data(), (size_t)c->B * 4);
    return G2048_OK;
}
int g2048_set_rng(g2048_ctx* c, const uint64_t* src) {
    if (!c || !src) return c ? fail(c, G2048_ERR_ARG, "null buffer") : G2048_ERR_ARG;
    for (uint32_t i = 0; i < c->B; ++i) c->rng[i] = Rng{src[2 * (size_t)i], src[2 * (size_t)i + 1]};
    return G2048_OK;
}
int g2048_get_rng(g2048_ctx* c, uint64_t* dst) {
    if (!c || !dst) return c ? fail(c, G2048_ERR_ARG, "null buffer") : G2048_ERR_ARG;
    for (uint32_t i = 0; i < c->B; ++i) {
        dst[2 * (size_t)i] = c->rng[i].s0;
        dst[2 * (size_t)i + 1] = c->rng[i].s1;
    }
    return G2048_OK;
}

int g2048_get_carry(g2048_ctx* c, uint8_t* prev, float* label, uint8_t* flags) {
    if (!c) return G2048_ERR_ARG;
    if (prev)
        for (uint32_t i = 0; i < c->B; ++i) store_board(prev + 16 * (size_t)i, c->prev[i]);
    if (label) memcpy(label, c->label.data(), (size_t)c->B * 4);
    if (flags) memcpy(flags, c->flags.data(), c->B);
    return G2048_OK;
}

int g2048_clear_carry(g2048_ctx* c) {
    if (!c) return G2048_ERR_ARG;
    std::fill(c->flags.begin(), c->flags.end(), (uint8_t)0);
    std::fill(c->label.begin(), c->label.end(), 0.0f);
    return G2048_OK;
}

int g2048_reset(g2048_ctx* c) {
    if (!c) return G2048_ERR_ARG;
    for (uint32_t i = 0; i < c->B; ++i) {
        c->boards[i] = new_game(c->rng[i]);
        c->scores[i] = 0;
        c->label[i] = 0.0f;
        c->flags[i] = 0;
    }
    if (c->log.lanes) log_init(c);
    return G2048_OK;
}

int g2048_set_auto_reset(g2048_ctx* c, int on) {
    if (!c) return G2048_ERR_ARG;
    c->auto_reset = on ? 1 : 0;
    return G2048_OK;
}

static void move_all_of(const Board& b, uint8_t* after, int32_t* reward, uint8_t* changed) {
    const Moves4 mv = all_moves(b);
    for (int d = 0; d < 4; ++d) {
        store_board(after + 16 * d, mv.m[d].after);
        reward[d] = (int32_t)merged_score(mv.m[d].ma, mv.m[d].mb);
    }
    *changed = (uint8_t)changed_mask(mv);
}

int g2048_move_all(g2048_ctx* c, uint8_t* after, int32_t* reward, uint8_t* changed) {
    if (!c || !after || !reward || !changed) return c ? fail(c, G2048_ERR_ARG, "null buffer") : G2048_ERR_ARG;
    for (uint32_t i = 0; i < c->B; ++i) move_all_of(c->boards[i], after + 64 * (size_t)i, reward + 4 * (size_t)i, changed + i);
    return G2048_OK;
}

int g2048_boards_move_all(g2048_ctx* c, const uint8_t* boards, int64_t count, uint8_t* after, int32_t* reward, uint8_t* changed) {
    if (!c || !boards || !after || !reward || !changed || count < 0) return c ? fail(c, G2048_ERR_ARG, "null buffer / bad count") : G2048_ERR_ARG;
    for (int64_t i = 0; i < count; ++i) move_all_of(load_board(boards + 16 * i), after + 64 * i, reward + 4 * i, changed + i);
    return G2048_OK;
}

int g2048_apply_moves(g2048_ctx* c, const uint8_t* dirs, uint8_t* moved) {
    if (!c || !dirs) return c ? fail(c, G2048_ERR_ARG, "null buffer") : G2048_ERR_ARG;
    for (uint32_t i = 0; i < c->B; ++i) {
        const Moves4 mv = all_moves(c->boards[i]);
        const Moved& m = mv.m[dirs[i] & 3u];
        c->boards[i] = m.after;
        c->scores[i] += (int32_t)merged_score(m.ma, m.mb);
        if (moved) moved[i] = m.changed;
    }
    return G2048_OK;
}

int g2048_terminal(g2048_ctx* c, uint8_t* over, uint8_t* n_empty, uint8_t* n_pairs) {
    if (!c) return G2048_ERR_ARG;
    for (uint32_t i = 0; i < c->B; ++i) {
        const Board& b = c->boards[i];
        if (over) over[i] = game_over(b);
        if (n_empty) n_empty[i] = (uint8_t)empty_count(b);
        if (n_pairs) n_pairs[i] = (uint8_t)adjacent_pairs(b);
    }
    return G2048_OK;
}

static int spawn_impl(g2048_ctx* c, const uint8_t* in_r10, const uint8_t* in_k, uint8_t* out_r10, uint8_t* out_k) {
    for (uint32_t i = 0; i < c->B; ++i) {
        Board& b = c->boards[i];
        const uint32_t e = empty_bits(b);
        uint32_t r10 = 255, k = 255;
        if (e) {
            if (in_r10) {
                r10 = in_r10[i];
                k = in_k[i];
                if (k >= popcount32(e)) k = popcount32(e) - 1;
            } else {
                spawn_draw(next_u64(c->rng[i]), popcount32(e), r10, k);
            }
            place_tile(b, r10, k, e);
        }
        if (out_r10) out_r10[i] = (uint8_t)r10;
        if (out_k) out_k[i] = (uint8_t)k;
    }
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

int g2048_step_random(g2048_ctx* c, uint32_t nsteps) {
    if (!c) return G2048_ERR_ARG;
    const int nthreads = c->threads;
    std::vector<g2048_stats> part((size_t)nthreads);
    for (auto& s : part) memset(&s, 0, sizeof s);
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (long long ii = 0; ii < (long long)c->B; ++ii) {
        const uint32_t i = (uint32_t)ii;
#ifdef _OPENMP
        g2048_stats& st = part[(size_t)omp_get_thread_num()];
#else
        g2048_stats& st = part[0];
#endif
        Board b = c->boards[i];
        Rng g = c->rng[i];
        int32_t score = c->scores[i];
        uint8_t fl = c->flags[i];
        for (uint32_t s = 0; s < nsteps; ++s) {
            if (fl & DONE) continue;
            const Moves4 mv = all_moves(b);
            const uint32_t mask = changed_mask(mv);
            bool over;
            if (mask) {
                const uint32_t j = pick_draw(next_u64(g), popcount32(mask));
                const Moved& m = mv.m[kth_set_bit(mask, j)];
                b = m.after;
                score += (int32_t)merged_score(m.ma, m.mb);
                st.moves += 1;
                st.valid_dirs += popcount32(mask);
                spawn(b, g);
                over = game_over(b) || max_tile(b) >= 16u;
            } else {
                over = true;
            }
            if (over) {
                count_finished(st, b, score, max_tile(b) >= 16u);
                if (c->auto_reset) {
                    b = new_game(g);
                    score = 0;
                } else {
                    fl |= DONE;
                }
            }
        }
        c->boards[i] = b;
        c->rng[i] = g;
        c->scores[i] = score;
        c->flags[i] = fl;
    }
    for (const auto& s : part) merge_stats(c->stats, s);
    return G2048_OK;
}

int g2048_features(g2048_ctx* c, int32_t* out) {
    if (!c || !out) return c ? fail(c, G2048_ERR_ARG, "null buffer") : G2048_ERR_ARG;
    NEED_TABLE(c);
    BY_N(c, features_impl<N>(c, out));
    return G2048_OK;
}

int g2048_weights_set(g2048_ctx* c, const float* w, int64_t count) {
    if (!c || !w) return c ? fail(c, G2048_ERR_ARG, "null buffer") : G2048_ERR_ARG;
    NEED_TABLE(c);
    NEED(c, count == (int64_t)c->slots, "weight count does not match the table");
    memcpy(c->table->w.data(), w, c->slots * 4);
    return G2048_OK;
}

int g2048_weights_get(g2048_ctx* c, float* w, int64_t count) {
    if (!c || !w) return c ? fail(c, G2048_ERR_ARG, "null buffer") : G2048_ERR_ARG;
    NEED_TABLE(c);
    NEED(c, count == (int64_t)c->slots, "weight count does not match the table");
    memcpy(w, c->table->w.data(), c->slots * 4);
    return G2048_OK;
}

// init_weights (r_learning.py:139-149): U[0, scale) per slot, counter-based (the statement of 2048_amd/rng.py:init_weights_np)
int g2048_weights_init(g2048_ctx* c, uint64_t seed, float scale) {
    if (!c) return G2048_ERR_ARG;
    NEED_TABLE(c);
    float* w = c->table->w.data();
#pragma omp parallel for num_threads(c->threads) schedule(static)
    for (long long i = 0; i < (long long)c->slots; ++i) {
        uint64_t x = seed + (uint64_t)i;
        const uint64_t z = splitmix64(x);
        w[i] = (float)(z >> 40) * (1.0f / 16777216.0f) * scale;
    }
    return G2048_OK;
}

int g2048_evaluate(g2048_ctx* c, float* value) {
    if (!c || !value) return c ? fail(c, G2048_ERR_ARG, "null buffer") : G2048_ERR_ARG;
    NEED_TABLE(c);
    BY_N(c, evaluate_impl<N>(c, c->boards.data(), nullptr, (int64_t)c->B, value));
    return G2048_OK;
}

int g2048_boards_evaluate(g2048_ctx* c, const uint8_t* boards, int64_t count, float* value) {
    if (!c || !boards || !value || count < 0) return c ? fail(c, G2048_ERR_ARG, "null buffer / bad count") : G2048_ERR_ARG;
    NEED_TABLE(c);
    BY_N(c, evaluate_impl<N>(c, nullptr, boards, count, value));
    return G2048_OK;
}

int g2048_boards_look_forward(g2048_ctx* c, const uint8_t* boards, int64_t count, int depth, int width, int since_empty, const uint64_t* salt, float* value) {
    if (!c || !boards || !value || count < 0) return c ? fail(c, G2048_ERR_ARG, "null buffer / bad count") : G2048_ERR_ARG;
    NEED_TABLE(c);
    if (int rc = la_args(c, depth, width, since_empty, 0)) return rc;
    const float* w = c->table->w.data();
    const int nthreads = c->threads;
    (void)nthreads;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
    for (long long i = 0; i < (long long)count; ++i) {
        const Salt s = salt ? Salt{salt[2 * i], salt[2 * i + 1]} : Salt{0, 0};
        const Board b = load_board(boards + 16 * i);
        switch (c->n) {
            case 2: value[i] = look_forward<2>(w, b, depth, width, since_empty, s); break;
            case 3: value[i] = look_forward<3>(w, b, depth, width, since_empty, s); break;
            case 4: value[i] = look_forward<4>(w, b, depth, width, since_empty, s); break;
            case 5: value[i] = look_forward<5>(w, b, depth, width, since_empty, s); break;
            default: value[i] = look_forward<6>(w, b, depth, width, since_empty, s); break;
        }
    }
    return G2048_OK;
}

int g2048_lookahead_steps(g2048_ctx* c, int depth, int width, int since_empty, int limit_tile, uint32_t nsteps) {
    if (!c) return G2048_ERR_ARG;
    NEED_TABLE(c);
    if (int rc = la_args(c, depth, width, since_empty, limit_tile)) return rc;
    for (uint32_t s = 0; s < nsteps; ++s) BY_N(c, lookahead_step<N>(c, depth, width, since_empty, (uint32_t)std::min(limit_tile, 255)));
    return G2048_OK;
}

int g2048_eval_select(g2048_ctx* c, float* value, uint8_t* action, float* values4) {
    if (!c || ((value == nullptr) != (action == nullptr))) return c ? fail(c, G2048_ERR_ARG, "null buffer") : G2048_ERR_ARG;
    NEED_TABLE(c);
    std::vector<float> scratch_v;
    std::vector<uint8_t> scratch_a;
    if (!value) {       // "device-only" run: compute, keep nothing
        scratch_v.resize(c->B);
        scratch_a.resize(c->B);
        value = scratch_v.data();
        action = scratch_a.data();
    }
    BY_N(c, eval_select_impl<N>(c, value, action, values4));
    return G2048_OK;
}

int g2048_update(g2048_ctx* c, const uint8_t* states, const float* dw, int64_t count) {
    if (!c || !states || !dw) return c ? fail(c, G2048_ERR_ARG, "null buffer") : G2048_ERR_ARG;
    NEED_TABLE(c);
    NEED(c, count >= 0 && count <= (1 << 28), "bad record count");
    BY_N(c, update_impl<N>(c, states, dw, count));
    return G2048_OK;
}

int g2048_td_steps(g2048_ctx* c, float alpha, uint32_t nsteps) {
    if (!c) return G2048_ERR_ARG;
    NEED_TABLE(c);
    try {
        for (uint32_t s = 0; s < nsteps; ++s) BY_N(c, td_step<N>(c, alpha));
    } catch (const std::bad_alloc&) {
        return fail(c, G2048_ERR_NOMEM, "out of host memory in a TD step");
    }
    return G2048_OK;
}

int g2048_set_lane_sort(g2048_ctx* c, uint32_t) { return c ? G2048_OK : G2048_ERR_ARG; }      // (a cache-line matter of the GPU: nothing to do)

int g2048_debug_lane_order(g2048_ctx* c, uint32_t*, uint16_t*) { return c ? fail(c, G2048_ERR_STATE, "no lane re-order on the CPU backend") : G2048_ERR_ARG; }

int g2048_set_update_mode(g2048_ctx* c, int mode) {
    if (!c) return G2048_ERR_ARG;
    NEED_TABLE(c);
    NEED(c, mode == 0 || mode == 1, "update mode must be 0 or 1");
    NEED(c, mode == 1 || c->update_rule == 0, "the per-slot mean rule needs update mode 1");
    c->update_mode = mode;      // (both modes form the same sums; here they are one code path)
    return G2048_OK;
}

int g2048_set_update_rule(g2048_ctx* c, int rule) {
    if (!c) return G2048_ERR_ARG;
    NEED_TABLE(c);
    NEED(c, rule == 0 || rule == 1, "update rule must be 0 (add every dw) or 1 (per-slot mean)");
    if (rule == 1 && c->update_mode != 1) return fail(c, G2048_ERR_ARG, "the per-slot mean rule needs the LDS-owner update (update mode 1)");
    c->update_rule = rule;
    return G2048_OK;
}

int g2048_td_steps_profiled(g2048_ctx* c, float alpha, uint32_t nsteps, float* ms_play, float* ms_update) {
    if (!c || !ms_play || !ms_update) return G2048_ERR_ARG;
    const auto t0 = std::chrono::steady_clock::now();
    if (int rc = g2048_td_steps(c, alpha, nsteps)) return rc;
    *ms_play = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count() / (float)(nsteps ? nsteps : 1);
    *ms_update = 0.0f;      // (one code path here: the whole step is reported as "play")
    return G2048_OK;
}

int g2048_td_steps_kernel_ms(g2048_ctx* c, float alpha, uint32_t nsteps, float* out4) {
    if (!c || !out4) return G2048_ERR_ARG;
    out4[1] = out4[2] = out4[3] = 0.0f;
    float upd;
    return g2048_td_steps_profiled(c, alpha, nsteps, out4, &upd);
}

int g2048_debug_owner_plan(g2048_ctx* c, uint64_t*, uint32_t, uint32_t* count) {
    if (!c || !count) return G2048_ERR_ARG;
    *count = 0;
    return G2048_OK;
}

int g2048_get_last_move(g2048_ctx* c, uint16_t* out) {
    if (!c || !out) return c ? fail(c, G2048_ERR_ARG, "null buffer") : G2048_ERR_ARG;
    memcpy(out, c->last_move.data(), (size_t)c->B * 2);
    return G2048_OK;
}

int g2048_log_enable(g2048_ctx* c, uint32_t lanes, uint32_t capacity) {
    if (!c) return G2048_ERR_ARG;
    NEED(c, lanes <= c->B && capacity <= (1u << 20) && (lanes == 0) == (capacity == 0), "bad log geometry");
    c->log = GameLogHost{};
    if (lanes == 0) return G2048_OK;
    try {
        c->log.moves.assign((size_t)lanes * 2 * capacity, 0);
        c->log.start.assign((size_t)lanes * 2, Board{{0, 0, 0, 0}});
        c->log.final.assign((size_t)lanes * 2, Board{{0, 0, 0, 0}});
        c->log.meta.assign((size_t)lanes * 8, 0u);
    } catch (const std::bad_alloc&) {
        c->log = GameLogHost{};
        return fail(c, G2048_ERR_NOMEM, "game log");
    }
    c->log.lanes = lanes;
    c->log.capacity = capacity;
    log_init(c);
    return G2048_OK;
}

int g2048_log_meta(g2048_ctx* c, uint32_t* out) {
    if (!c || !out) return c ? fail(c, G2048_ERR_ARG, "null buffer") : G2048_ERR_ARG;
    if (!c->log.lanes) return fail(c, G2048_ERR_STATE, "game log is not enabled");
    memcpy(out, c->log.meta.data(), (size_t)c->log.lanes * 8 * 4);
    return G2048_OK;
}

int g2048_log_game(g2048_ctx* c, uint32_t lane, uint32_t slot, uint16_t* moves, uint8_t* start) {
    if (!c || !moves || !start) return c ? fail(c, G2048_ERR_ARG, "null buffer") : G2048_ERR_ARG;
    if (!c->log.lanes) return fail(c, G2048_ERR_STATE, "game log is not enabled");
    NEED(c, lane < c->log.lanes && slot < 2, "bad lane / slot");
    memcpy(moves, c->log.moves.data() + ((size_t)lane * 2 + slot) * c->log.capacity, (size_t)c->log.capacity * 2);
    store_board(start, c->log.start[(size_t)lane * 2 + slot]);
    return G2048_OK;
}

int g2048_log_final(g2048_ctx* c, uint32_t lane, uint32_t slot, uint8_t* board) {
    if (!c || !board) return c ? fail(c, G2048_ERR_ARG, "null buffer") : G2048_ERR_ARG;
    if (!c->log.lanes) return fail(c, G2048_ERR_STATE, "game log is not enabled");
    NEED(c, lane < c->log.lanes && slot < 2, "bad lane / slot");
    store_board(board, c->log.final[(size_t)lane * 2 + slot]);
    return G2048_OK;
}

int g2048_stats_get(g2048_ctx* c, g2048_stats* out) {
    if (!c || !out) return G2048_ERR_ARG;
    *out = c->stats;
    return G2048_OK;
}

int g2048_stats_reset(g2048_ctx* c) {
    if (!c) return G2048_ERR_ARG;
    memset(&c->stats, 0, sizeof c->stats);
    return G2048_OK;
}

// ---- multi-GPU group.  "Device" pointers are host pointers on this backend.
int g2048_weights_device_ptr(g2048_ctx* c, void** ptr, int64_t* count) {
    if (!c || !ptr) return G2048_ERR_ARG;
    NEED_TABLE(c);
    *ptr = c->table->w.data();
    if (count) *count = (int64_t)c->slots;
    return G2048_OK;
}

int g2048_delta_begin(g2048_ctx* c) {
    if (!c) return G2048_ERR_ARG;
    NEED_TABLE(c);
    c->table->w0 = c->table->w;
    c->table->tracking = true;
    return G2048_OK;
}

static float* own_delta(g2048_ctx* c) {
    Table& t = *c->table;
    if (t.delta.size() != c->slots) t.delta.assign(c->slots, 0.0f);
    return t.delta.data();
}

int g2048_delta_extract(g2048_ctx* c, void* dst) {        // dst == NULL: into the context's own buffer
    if (!c) return G2048_ERR_ARG;
    NEED_TABLE(c);
    if (!c->table->tracking) return fail(c, G2048_ERR_STATE, "g2048_delta_begin was not called");
    float* d = dst ? (float*)dst : own_delta(c);
    const Table& t = *c->table;
    for (size_t i = 0; i < c->slots; ++i) d[i] = t.w[i] - t.w0[i];
    return G2048_OK;
}

int g2048_delta_apply(g2048_ctx* c, const void* src) {    // src == NULL: from the context's own buffer
    if (!c) return G2048_ERR_ARG;
    NEED_TABLE(c);
    if (!c->table->tracking) return fail(c, G2048_ERR_STATE, "g2048_delta_begin was not called");
    const float* d = src ? (const float*)src : own_delta(c);
    Table& t = *c->table;
    for (size_t i = 0; i < c->slots; ++i) t.w0[i] = t.w[i] = t.w0[i] + d[i];
    return G2048_OK;
}

int g2048_delta_pack_touched(g2048_ctx* c, void* pack) {
    if (!c || !pack) return c ? fail(c, G2048_ERR_ARG, "null buffer") : G2048_ERR_ARG;
    NEED_TABLE(c);
    if (!c->table->tracking) return fail(c, G2048_ERR_STATE, "g2048_delta_begin was not called");
    float* p = (float*)pack;
    const Table& t = *c->table;
    for (size_t i = 0; i < c->slots; ++i) {
        const float d = t.w[i] - t.w0[i];
        p[i] = d;
        p[c->slots + i] = d != 0.0f ? 1.0f : 0.0f;
    }
    return G2048_OK;
}

int g2048_delta_apply_mean(g2048_ctx* c, const void* pack) {
    if (!c || !pack) return c ? fail(c, G2048_ERR_ARG, "null buffer") : G2048_ERR_ARG;
    NEED_TABLE(c);
    if (!c->table->tracking) return fail(c, G2048_ERR_STATE, "g2048_delta_begin was not called");
    const float* p = (const float*)pack;
    Table& t = *c->table;
    for (size_t i = 0; i < c->slots; ++i) {
        const float k = p[c->slots + i];
        t.w0[i] = t.w[i] = t.w0[i] + p[i] / (k > 1.0f ? k : 1.0f);
    }
    return G2048_OK;
}

int g2048_delta_device_ptr(g2048_ctx* c, void** ptr) {
    if (!c || !ptr) return G2048_ERR_ARG;
    NEED_TABLE(c);
    *ptr = own_delta(c);
    return G2048_OK;
}

int g2048_stream_handle(g2048_ctx* c, void** s) {
    if (!c || !s) return G2048_ERR_ARG;
    *s = nullptr;
    return G2048_OK;
}

int g2048_comm_unique_id(uint8_t*) { return G2048_ERR_COMM; }
int g2048_comm_init(g2048_ctx* c, int, int, const uint8_t*) { return c ? fail(c, G2048_ERR_COMM, "no RCCL in the CPU backend") : G2048_ERR_ARG; }
int g2048_comm_destroy(g2048_ctx* c) { return c ? G2048_OK : G2048_ERR_ARG; }
int g2048_comm_info(g2048_ctx* c, int* rank, int* nranks) {
    if (!c || !rank || !nranks) return G2048_ERR_ARG;
    *rank = 0;
    *nranks = 1;
    return G2048_OK;
}
int g2048_allreduce_deltas(g2048_ctx* c) { return c ? fail(c, G2048_ERR_COMM, "no RCCL in the CPU backend") : G2048_ERR_ARG; }
int g2048_allreduce_f64(g2048_ctx* c, double*, int, int) { return c ? fail(c, G2048_ERR_COMM, "no RCCL in the CPU backend") : G2048_ERR_ARG; }

int g2048_debug_phases(unsigned long long*, unsigned long long*, int) { return G2048_ERR_STATE; }
int g2048_debug_phases_hw(unsigned long long*, unsigned long long*) { return G2048_ERR_STATE; }

}  // extern "C"
