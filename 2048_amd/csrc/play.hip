// play.hip — the greedy afterstate choice (r_learning.py:229-237) and the kernels built on it: k_td_play* (step part 1 of
// QAgent.episode for every live lane) and k_eval_select* (BASELINE config 3).  See g2048.hip for the step's other half.
#include <hip/hip_runtime.h>
#include <stdio.h>

#include <algorithm>
#include <mutex>
#include <vector>

#include "play.hpp"

using namespace g2048;

namespace {

// how many candidate directions have their table gathers in flight together (register pressure against latency hiding)
#ifndef G2048_BATCH4_MAXF
#define G2048_BATCH4_MAXF 24
#endif
#ifndef G2048_BATCH2_MAXF
#define G2048_BATCH2_MAXF 40
#endif

struct Choice {
    int action;         // -1: no direction changes the board
    float value;
    float v[4];
};

// greedy afterstate choice (r_learning.py:229-237): strict '>' from -inf keeps the first maximum.
// The table reads of ALL candidate directions are issued before any of them is summed: the kernel is bound by the
// latency and rate of these 4-byte gathers (up to 4 x F per board, 64 separate lines per wave instruction), and one
// round of 4 F loads in flight per lane hides far more of it than four rounds of F.  Directions that do not change the
// board read slot 0 and are ignored (a few wasted loads of a hot line instead of a branch around every group).
template <int N>
__device__ __forceinline__ Choice choose(const float* __restrict__ w, const Moves4& mv) {
    constexpr int F = Shape<N>::F;
    Choice c;
    c.action = -1;
    c.value = -INFINITY;
    int first_valid = -1;
    if constexpr (F <= G2048_BATCH4_MAXF) {
        uint32_t s0[F], s1[F], s2[F], s3[F];
        memory_slots<N>(pack_pairs(mv.m0.after), s0);
        memory_slots<N>(pack_pairs(mv.m1.after), s1);
        memory_slots<N>(pack_pairs(mv.m2.after), s2);
        memory_slots<N>(pack_pairs(mv.m3.after), s3);
        float x0[F], x1[F], x2[F], x3[F];
#pragma unroll
        for (int f = 0; f < F; ++f) {
            x0[f] = ld_w(w, mv.m0.changed ? s0[f] : 0u);
            x1[f] = ld_w(w, mv.m1.changed ? s1[f] : 0u);
            x2[f] = ld_w(w, mv.m2.changed ? s2[f] : 0u);
            x3[f] = ld_w(w, mv.m3.changed ? s3[f] : 0u);
        }
        float v0 = 0.0f, v1 = 0.0f, v2 = 0.0f, v3 = 0.0f;      // each a left-to-right sum, as QAgent.evaluate
#pragma unroll
        for (int f = 0; f < F; ++f) {
            v0 += x0[f];
            v1 += x1[f];
            v2 += x2[f];
            v3 += x3[f];
        }
        c.v[0] = mv.m0.changed ? v0 : -INFINITY;
        c.v[1] = mv.m1.changed ? v1 : -INFINITY;
        c.v[2] = mv.m2.changed ? v2 : -INFINITY;
        c.v[3] = mv.m3.changed ? v3 : -INFINITY;
        const bool ch[4] = {mv.m0.changed, mv.m1.changed, mv.m2.changed, mv.m3.changed};
#pragma unroll
        for (int d = 0; d < 4; ++d)
            if (ch[d]) {
                if (first_valid < 0) first_valid = d;
                if (c.v[d] > c.value) {
                    c.value = c.v[d];
                    c.action = d;
                }
            }
    } else if constexpr (F <= G2048_BATCH2_MAXF) {
        // two directions per round (2 F loads in flight per lane)
#define G2048_TRY_PAIR(DA, MA, DB, MB)                                               \
    {                                                                                \
        uint32_t sa[F], sb[F];                                                       \
        memory_slots<N>(pack_pairs((MA).after), sa);                                \
        memory_slots<N>(pack_pairs((MB).after), sb);                                \
        float xa[F], xb[F];                                                          \
        _Pragma("unroll") for (int f = 0; f < F; ++f) {                              \
            xa[f] = ld_w_f<N>(w, (MA).changed ? sa[f] : 0u, f);                      \
            xb[f] = ld_w_f<N>(w, (MB).changed ? sb[f] : 0u, f);                      \
        }                                                                            \
        float va = 0.0f, vb = 0.0f;                                                  \
        _Pragma("unroll") for (int f = 0; f < F; ++f) {                              \
            va += xa[f];                                                             \
            vb += xb[f];                                                             \
        }                                                                            \
        c.v[DA] = (MA).changed ? va : -INFINITY;                                     \
        c.v[DB] = (MB).changed ? vb : -INFINITY;                                     \
        if ((MA).changed) {                                                          \
            if (first_valid < 0) first_valid = DA;                                   \
            if (va > c.value) { c.value = va; c.action = DA; }                       \
        }                                                                            \
        if ((MB).changed) {                                                          \
            if (first_valid < 0) first_valid = DB;                                   \
            if (vb > c.value) { c.value = vb; c.action = DB; }                       \
        }                                                                            \
    }
        G2048_TRY_PAIR(0, mv.m0, 1, mv.m1)
        G2048_TRY_PAIR(2, mv.m2, 3, mv.m3)
#undef G2048_TRY_PAIR
    } else {
#define G2048_TRY_DIR(D, M)                                  \
    c.v[D] = -INFINITY;                                      \
    if ((M).changed) {                                       \
        if (first_valid < 0) first_valid = D;                \
        float v = value_of<N>(w, (M).after);                 \
        c.v[D] = v;                                          \
        if (v > c.value) {                                   \
            c.value = v;                                     \
            c.action = D;                                    \
        }                                                    \
    }
        G2048_TRY_DIR(0, mv.m0)
        G2048_TRY_DIR(1, mv.m1)
        G2048_TRY_DIR(2, mv.m2)
        G2048_TRY_DIR(3, mv.m3)
#undef G2048_TRY_DIR
    }
    if (c.action < 0 && first_valid >= 0) {     // every value was -inf or NaN (a poisoned table): still make a legal move
        c.action = first_valid;
        c.value = first_valid == 0 ? c.v[0] : first_valid == 1 ? c.v[1] : first_valid == 2 ? c.v[2] : c.v[3];      // (no dynamic index: it would put v[] in scratch or LDS)
    }
    return c;
}

// ---- LDS-resident hot set of the table (k_td_play).  The gathers are bound by the CU's L1: a wave instruction touches
// ~57 separate cache lines, a third of them miss (TCP counters, profiles/), and the address path stalls behind them.
// Two thirds of a fresh agent's four-cell reads — half of a trained agent's — fall on tuples whose four tiles are all
// <= 32 (6^4 = 1 296 of a feature's 65 536 entries).  Those 17 x 1 296 entries (88 KB) are copied into LDS once per
// launch and read from there; the LDS takes divergent lanes about ten times faster than the L1's tag path, and the L1
// keeps its lines for the cold tail.
#ifndef G2048_HOT_ENTRIES        // entries per four-cell table kept in LDS: the first ones in the table's memory order (a power of two)
#define G2048_HOT_ENTRIES 2048
#endif
constexpr uint32_t HOT_PER_FEATURE = G2048_HOT_ENTRIES, HOT_FEATURES = 17u, HOT_SLOTS = HOT_FEATURES * HOT_PER_FEATURE;
// n = 2, 3 (round 3): n = 2's whole table is 24 KB and lives in LDS; n = 3 keeps the 8^3 = 512 entries of each of its 52
// tables whose three cells are all below 8 (tiles up to 128: every gather of a young board, most of a trained agent's) — 104 KB.
constexpr uint32_t SMALL3_PER_FEATURE = 512u;
template <int N> struct HotShape { static constexpr uint32_t SLOTS = HOT_SLOTS; };
template <> struct HotShape<2> { static constexpr uint32_t SLOTS = Shape<2>::SLOTS; };
template <> struct HotShape<3> { static constexpr uint32_t SLOTS = 52u * SMALL3_PER_FEATURE; };
// index of a three-cell table entry (a << 8 | b << 4 | c) in the LDS copy, if a, b, c < 8
__device__ __forceinline__ uint32_t small3_index(uint32_t rel) { return ((rel >> 2) & 0x1C0u) | ((rel >> 1) & 0x38u) | (rel & 7u); }

// In the table's memory order (table_place) the entries of small tiles come first: transposed index t < 256 = every cell of
// the tuple empty / 2 / 4 / 8, t < 1 024 = 45 % of a fresh agent's gathers and 33 % of a trained agent's
// (tools/hot_coverage.py).  So "hot" is one compare on a number the gather computes anyway and t is the LDS index.
template <int N, int TPB>
__device__ __forceinline__ void load_hot_set(float* hot, const float* __restrict__ w) {
    static_assert(HOT_PER_FEATURE % 4u == 0u, "the copy moves 16 bytes per thread and turn");
    for (uint32_t j = 4u * threadIdx.x; j < HotShape<N>::SLOTS; j += 4u * TPB) {
        if constexpr (N == 2) {
            *reinterpret_cast<float4*>(hot + j) = *reinterpret_cast<const float4*>(w + j);
        } else if constexpr (N == 3) {          // j = f * 512 + (a << 6 | b << 3 | c), c a multiple of 4: four consecutive table entries
            const uint32_t f = j / SMALL3_PER_FEATURE, t = j - f * SMALL3_PER_FEATURE;
            *reinterpret_cast<float4*>(hot + j) = *reinterpret_cast<const float4*>(w + f * 4096u + ((t >> 6) << 8 | ((t >> 3) & 7u) << 4 | (t & 7u)));
        } else {
            const uint32_t f = j / HOT_PER_FEATURE, t = j - f * HOT_PER_FEATURE;
            *reinterpret_cast<float4*>(hot + j) = *reinterpret_cast<const float4*>(w + f * 65536u + t);
        }
    }
    __syncthreads();
}

// choose<N> with the hot set (n = 4, 5: features 0..16 are the four-cell tuples), in two phases so that no gather needs a
// second result register and none waits for another: (A) EVERY lane reads the LDS copy at t mod HOT (a wrong word for the
// cold lanes, at LDS speed); (B) the cold lanes then overwrite it with a global load under their exec mask.  The L1 sees
// the cold lanes only.  (Round 2's first version — every cell <= 32 through a base-6 index, ONE flat load per gather whose
// lanes point into either aperture — removed 44 % of the L1's tag look-ups and no time: a flat load still walks every lane
// through the address unit and the tag stage.  A ds_read / global_load pair under complementary exec masks, LDS second,
// makes the compiler wait for each global load before it issues the ds_read into the same register.)
//
// The 68 results live in VGPRs THE COMPILER DOES NOT HAVE (round 3).  Round 2 issued the masked loads from inline asm into
// ordinary variables ("+v"(x[f])): the compiler believed x[f] valid from that point on while the load was still in flight
// until a fence further down, and was free to copy or spill it in between — tools/check_codeobj.py found it doing exactly
// that in the shipped k_td_play<5, 512, true, true> (v_mov_b32 v209, v2 some 1 900 instructions after global_load_dword v2,
// no wait in between: right only because the load had long returned).  Now the kernels that use this path carry
// amdgpu_waves_per_eu(3, 3): the register allocator is then held to the budget of three waves per SIMD, 512 / 3 -> 168 VGPRs
// (v0 .. v167; it spills rather than exceed it — amdgpu_num_vgpr, the attribute made for this, is silently ignored on gfx950:
// tools/check_codeobj.py showed the compiler at home in the "reserved" range), and v168 .. v235 are named only inside the asm
// statements below — the LDS read, the masked global load, and, behind `s_waitcnt vmcnt(0)`, the v_add_f32 that consumes
// each of them.  (The clobber of the last one puts them into the kernel's VGPR count; the kernel still runs two waves per
// SIMD, one 512-thread workgroup per CU, as its LDS footprint dictates.)  Nothing the compiler generates can touch a result
// in flight, and tools/check_codeobj.py (run by __graft_entry__.build()) checks the code object for it: no scratch, no
// instruction that names a load's destination before a wait that covers it.
#define G2048_HOT_REGS0(M) M(0, "v168") M(1, "v169") M(2, "v170") M(3, "v171") M(4, "v172") M(5, "v173") M(6, "v174") M(7, "v175") M(8, "v176") M(9, "v177") M(10, "v178") M(11, "v179") M(12, "v180") M(13, "v181") M(14, "v182") M(15, "v183") M(16, "v184")
#define G2048_HOT_REGS1(M) M(0, "v185") M(1, "v186") M(2, "v187") M(3, "v188") M(4, "v189") M(5, "v190") M(6, "v191") M(7, "v192") M(8, "v193") M(9, "v194") M(10, "v195") M(11, "v196") M(12, "v197") M(13, "v198") M(14, "v199") M(15, "v200") M(16, "v201")
#define G2048_HOT_REGS2(M) M(0, "v202") M(1, "v203") M(2, "v204") M(3, "v205") M(4, "v206") M(5, "v207") M(6, "v208") M(7, "v209") M(8, "v210") M(9, "v211") M(10, "v212") M(11, "v213") M(12, "v214") M(13, "v215") M(14, "v216") M(15, "v217") M(16, "v218")
#define G2048_HOT_REGS3(M) M(0, "v219") M(1, "v220") M(2, "v221") M(3, "v222") M(4, "v223") M(5, "v224") M(6, "v225") M(7, "v226") M(8, "v227") M(9, "v228") M(10, "v229") M(11, "v230") M(12, "v231") M(13, "v232") M(14, "v233") M(15, "v234") M(16, "v235")
#define PLAY_HOT_WAVES_PER_EU 3           // the allocator owns 512 / 3 -> 168 VGPRs: v0 .. v167
#define PLAY_HOT_LAST_VGPR "v235"

template <int N>
__device__ __forceinline__ Choice choose_hot(const float* __restrict__ w, const float* hot_, const Moves4& mv) {
    constexpr int F = Shape<N>::F;
    static_assert(N == 4 || N == 5, "the hot set covers the four-cell features; all four directions' gathers are in flight together");
    // (`hot_` IS the kernel's __shared__ array, which the compiler cannot see through the lambda it arrives by)
    const uint32_t hot_base = (uint32_t)(size_t)(const __attribute__((address_space(3))) float*)hot_;
    asm volatile("" ::: PLAY_HOT_LAST_VGPR);        // (counts the registers above the allocator's range into the kernel's VGPR budget)
    Choice c;
    c.action = -1;
    c.value = -INFINITY;
    int first_valid = -1;
    // phase A + B of one direction: T[f] = the gather's place in its table (memory order), hot iff T[f] < HOT_PER_FEATURE
#define G2048_HOT_LDS(f, reg) asm volatile("ds_read_b32 " reg ", %0" ::"v"(hot_base + (((uint32_t)(f) * HOT_PER_FEATURE + (T_[f] & (HOT_PER_FEATURE - 1u))) << 2)));
#define G2048_HOT_COLD(f) (T_[f] >= HOT_PER_FEATURE)
#define G2048_HOT_GLB(f, reg) \
    if (changed_ && G2048_HOT_COLD(f)) asm volatile("global_load_dword " reg ", %0, %1" ::"v"(((uint32_t)(f) * 65536u + T_[f]) << 2), "s"(w));
#define G2048_HOT_DIR(M, REGS, XC)                                                          \
    float XC[F > 17 ? F - 17 : 1];                                                          \
    {                                                                                       \
        uint32_t ms_[F], T_[17];                                                            \
        const bool changed_ = (M).changed;                                                  \
        memory_slots<N>(pack_pairs((M).after), ms_);                                        \
        _Pragma("unroll") for (int f = 0; f < 17; ++f) T_[f] = ms_[f] - (uint32_t)f * 65536u;      /* the place inside the feature's table */ \
        REGS(G2048_HOT_LDS)                                                                 \
        asm volatile("s_waitcnt lgkmcnt(0)");       /* the LDS words are in before a global load may land on top of them */ \
        REGS(G2048_HOT_GLB)                                                                 \
        _Pragma("unroll") for (int f = 17; f < F; ++f) XC[f - 17] = ld_w(w, changed_ ? ms_[f] : 0u); \
    }
#define G2048_HOT_ADD(f, reg) asm volatile("v_add_f32 %0, %0, " reg : "+v"(acc_));
#define G2048_HOT_SUM(REGS, XC, V)          /* a left-to-right sum from 0, as QAgent.evaluate */ \
    float V;                                                                                \
    {                                                                                       \
        float acc_ = 0.0f;                                                                  \
        REGS(G2048_HOT_ADD)                                                                 \
        _Pragma("unroll") for (int f = 17; f < F; ++f) acc_ += XC[f - 17];                 \
        V = acc_;                                                                           \
    }
    G2048_HOT_DIR(mv.m0, G2048_HOT_REGS0, xc0)
    G2048_HOT_DIR(mv.m1, G2048_HOT_REGS1, xc1)
    G2048_HOT_DIR(mv.m2, G2048_HOT_REGS2, xc2)
    G2048_HOT_DIR(mv.m3, G2048_HOT_REGS3, xc3)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    G2048_HOT_SUM(G2048_HOT_REGS0, xc0, v0)
    G2048_HOT_SUM(G2048_HOT_REGS1, xc1, v1)
    G2048_HOT_SUM(G2048_HOT_REGS2, xc2, v2)
    G2048_HOT_SUM(G2048_HOT_REGS3, xc3, v3)
    c.v[0] = mv.m0.changed ? v0 : -INFINITY;
    c.v[1] = mv.m1.changed ? v1 : -INFINITY;
    c.v[2] = mv.m2.changed ? v2 : -INFINITY;
    c.v[3] = mv.m3.changed ? v3 : -INFINITY;
#undef G2048_HOT_LDS
#undef G2048_HOT_GLB
#undef G2048_HOT_DIR
#undef G2048_HOT_ADD
#undef G2048_HOT_SUM
    const bool ch[4] = {mv.m0.changed, mv.m1.changed, mv.m2.changed, mv.m3.changed};
#pragma unroll
    for (int d = 0; d < 4; ++d)
        if (ch[d]) {
            if (first_valid < 0) first_valid = d;
            if (c.v[d] > c.value) {
                c.value = c.v[d];
                c.action = d;
            }
        }
    if (c.action < 0 && first_valid >= 0) {     // every value was -inf or NaN (a poisoned table): still make a legal move
        c.action = first_valid;
        c.value = first_valid == 0 ? c.v[0] : first_valid == 1 ? c.v[1] : first_valid == 2 ? c.v[2] : c.v[3];      // (no dynamic index: it would put v[] in scratch or LDS)
    }
    return c;
}

// choose<N> for n = 2, 3 with the LDS copy (HotShape<N>).
// n = 2: every gather is an LDS read.  n = 3: the two phases of choose_hot — (A) every lane reads the LDS copy, (B) the lanes
// with a cell >= 8 overwrite the word with a global load under their exec mask — two directions (2 x 52 gathers) in flight
// at a time, the results in VGPRs above the register allocator's range (amdgpu_waves_per_eu(4, 4): v0 .. v127 are the
// compiler's, v128 .. v231 the gathers'; see choose_hot).
#define G2048_S3_REGS0(M) M(0, "v128") M(1, "v129") M(2, "v130") M(3, "v131") M(4, "v132") M(5, "v133") M(6, "v134") M(7, "v135") M(8, "v136") M(9, "v137") M(10, "v138") M(11, "v139") M(12, "v140") M(13, "v141") M(14, "v142") M(15, "v143") M(16, "v144") M(17, "v145") M(18, "v146") M(19, "v147") M(20, "v148") M(21, "v149") M(22, "v150") M(23, "v151") M(24, "v152") M(25, "v153") M(26, "v154") M(27, "v155") M(28, "v156") M(29, "v157") M(30, "v158") M(31, "v159") M(32, "v160") M(33, "v161") M(34, "v162") M(35, "v163") M(36, "v164") M(37, "v165") M(38, "v166") M(39, "v167") M(40, "v168") M(41, "v169") M(42, "v170") M(43, "v171") M(44, "v172") M(45, "v173") M(46, "v174") M(47, "v175") M(48, "v176") M(49, "v177") M(50, "v178") M(51, "v179")
#define G2048_S3_REGS1(M) M(0, "v180") M(1, "v181") M(2, "v182") M(3, "v183") M(4, "v184") M(5, "v185") M(6, "v186") M(7, "v187") M(8, "v188") M(9, "v189") M(10, "v190") M(11, "v191") M(12, "v192") M(13, "v193") M(14, "v194") M(15, "v195") M(16, "v196") M(17, "v197") M(18, "v198") M(19, "v199") M(20, "v200") M(21, "v201") M(22, "v202") M(23, "v203") M(24, "v204") M(25, "v205") M(26, "v206") M(27, "v207") M(28, "v208") M(29, "v209") M(30, "v210") M(31, "v211") M(32, "v212") M(33, "v213") M(34, "v214") M(35, "v215") M(36, "v216") M(37, "v217") M(38, "v218") M(39, "v219") M(40, "v220") M(41, "v221") M(42, "v222") M(43, "v223") M(44, "v224") M(45, "v225") M(46, "v226") M(47, "v227") M(48, "v228") M(49, "v229") M(50, "v230") M(51, "v231")
#define PLAY_SMALL3_WAVES_PER_EU 4        // k_eval_select_lds3: the allocator owns 512 / 4 = 128 VGPRs (v0 .. v127), two directions in flight
#define PLAY_SMALL3_LAST_VGPR "v231"
// k_td_play_lds3 needs more than 128 registers for everything else a step does: budget of three waves (v0 .. v167), ONE
// direction's 52 gathers in flight at a time
#define G2048_S3P_REGS(M) M(0, "v168") M(1, "v169") M(2, "v170") M(3, "v171") M(4, "v172") M(5, "v173") M(6, "v174") M(7, "v175") M(8, "v176") M(9, "v177") M(10, "v178") M(11, "v179") M(12, "v180") M(13, "v181") M(14, "v182") M(15, "v183") M(16, "v184") M(17, "v185") M(18, "v186") M(19, "v187") M(20, "v188") M(21, "v189") M(22, "v190") M(23, "v191") M(24, "v192") M(25, "v193") M(26, "v194") M(27, "v195") M(28, "v196") M(29, "v197") M(30, "v198") M(31, "v199") M(32, "v200") M(33, "v201") M(34, "v202") M(35, "v203") M(36, "v204") M(37, "v205") M(38, "v206") M(39, "v207") M(40, "v208") M(41, "v209") M(42, "v210") M(43, "v211") M(44, "v212") M(45, "v213") M(46, "v214") M(47, "v215") M(48, "v216") M(49, "v217") M(50, "v218") M(51, "v219")
#define PLAY_SMALL3P_WAVES_PER_EU 3
#define PLAY_SMALL3P_LAST_VGPR "v219"

template <int N, bool PAIRS = true>
__device__ __forceinline__ Choice choose_small(const float* __restrict__ w, const float* hot_, const Moves4& mv) {
    constexpr int F = Shape<N>::F;
    static_assert(N == 2 || N == 3, "n = 2, 3");
    Choice c;
    c.action = -1;
    c.value = -INFINITY;
    int first_valid = -1;
    if constexpr (N == 2) {
        const __attribute__((address_space(3))) float* hot = (const __attribute__((address_space(3))) float*)hot_;
        uint32_t s0[F], s1[F], s2[F], s3[F];
        feature_slots<N>(pack_board(mv.m0.after), s0);
        feature_slots<N>(pack_board(mv.m1.after), s1);
        feature_slots<N>(pack_board(mv.m2.after), s2);
        feature_slots<N>(pack_board(mv.m3.after), s3);
        float x0[F], x1[F], x2[F], x3[F];
#pragma unroll
        for (int f = 0; f < F; ++f) {
            x0[f] = hot[s0[f]];
            x1[f] = hot[s1[f]];
            x2[f] = hot[s2[f]];
            x3[f] = hot[s3[f]];
        }
        float v0 = 0.0f, v1 = 0.0f, v2 = 0.0f, v3 = 0.0f;      // each a left-to-right sum, as QAgent.evaluate
#pragma unroll
        for (int f = 0; f < F; ++f) {
            v0 += x0[f];
            v1 += x1[f];
            v2 += x2[f];
            v3 += x3[f];
        }
        c.v[0] = mv.m0.changed ? v0 : -INFINITY;
        c.v[1] = mv.m1.changed ? v1 : -INFINITY;
        c.v[2] = mv.m2.changed ? v2 : -INFINITY;
        c.v[3] = mv.m3.changed ? v3 : -INFINITY;
    } else {
        const uint32_t hot_base = (uint32_t)(size_t)(const __attribute__((address_space(3))) float*)hot_;
        // (counts the registers above the allocator's range into the kernel's VGPR budget)
        if constexpr (PAIRS)
            asm volatile("" ::: PLAY_SMALL3_LAST_VGPR);
        else
            asm volatile("" ::: PLAY_SMALL3P_LAST_VGPR);
#define G2048_S3_LDS(f, reg) \
    asm volatile("ds_read_b32 " reg ", %0" ::"v"(hot_base + (((uint32_t)(f) * SMALL3_PER_FEATURE + small3_index(s_[f] - (uint32_t)(f) * 4096u)) << 2)));
#define G2048_S3_GLB(f, reg) \
    if (changed_ && (s_[f] & 0x888u)) asm volatile("global_load_dword " reg ", %0, %1" ::"v"(s_[f] << 2), "s"(w));
#define G2048_S3_DIR(M, REGS)                                                               \
    {                                                                                       \
        uint32_t s_[F];                                                                     \
        const bool changed_ = (M).changed;                                                  \
        feature_slots<N>(pack_board((M).after), s_);        /* (feature f's slots start at f * 4096: bits 3, 7, 11 are the cells' bit 3) */ \
        REGS(G2048_S3_LDS)                                                                  \
        asm volatile("s_waitcnt lgkmcnt(0)");       /* the LDS words are in before a global load may land on top of them */ \
        REGS(G2048_S3_GLB)                                                                  \
    }
#define G2048_S3_ADD(f, reg) asm volatile("v_add_f32 %0, %0, " reg : "+v"(acc_));
#define G2048_S3_SUM(REGS, V)               /* a left-to-right sum from 0, as QAgent.evaluate */ \
    float V;                                                                                \
    {                                                                                       \
        float acc_ = 0.0f;                                                                  \
        REGS(G2048_S3_ADD)                                                                  \
        V = acc_;                                                                           \
    }
        float v0, v1, v2, v3;
        if constexpr (PAIRS) {
            G2048_S3_DIR(mv.m0, G2048_S3_REGS0)
            G2048_S3_DIR(mv.m1, G2048_S3_REGS1)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            G2048_S3_SUM(G2048_S3_REGS0, a0)
            G2048_S3_SUM(G2048_S3_REGS1, a1)
            G2048_S3_DIR(mv.m2, G2048_S3_REGS0)
            G2048_S3_DIR(mv.m3, G2048_S3_REGS1)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            G2048_S3_SUM(G2048_S3_REGS0, a2)
            G2048_S3_SUM(G2048_S3_REGS1, a3)
            v0 = a0; v1 = a1; v2 = a2; v3 = a3;
        } else {
            G2048_S3_DIR(mv.m0, G2048_S3P_REGS)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            G2048_S3_SUM(G2048_S3P_REGS, a0)
            G2048_S3_DIR(mv.m1, G2048_S3P_REGS)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            G2048_S3_SUM(G2048_S3P_REGS, a1)
            G2048_S3_DIR(mv.m2, G2048_S3P_REGS)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            G2048_S3_SUM(G2048_S3P_REGS, a2)
            G2048_S3_DIR(mv.m3, G2048_S3P_REGS)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            G2048_S3_SUM(G2048_S3P_REGS, a3)
            v0 = a0; v1 = a1; v2 = a2; v3 = a3;
        }
#undef G2048_S3_LDS
#undef G2048_S3_GLB
#undef G2048_S3_DIR
#undef G2048_S3_ADD
#undef G2048_S3_SUM
        c.v[0] = mv.m0.changed ? v0 : -INFINITY;
        c.v[1] = mv.m1.changed ? v1 : -INFINITY;
        c.v[2] = mv.m2.changed ? v2 : -INFINITY;
        c.v[3] = mv.m3.changed ? v3 : -INFINITY;
    }
    const bool ch[4] = {mv.m0.changed, mv.m1.changed, mv.m2.changed, mv.m3.changed};
#pragma unroll
    for (int d = 0; d < 4; ++d)
        if (ch[d]) {
            if (first_valid < 0) first_valid = d;
            if (c.v[d] > c.value) {
                c.value = c.v[d];
                c.action = d;
            }
        }
    if (c.action < 0 && first_valid >= 0) {     // every value was -inf or NaN (a poisoned table): still make a legal move
        c.action = first_valid;
        c.value = first_valid == 0 ? c.v[0] : first_valid == 1 ? c.v[1] : first_valid == 2 ? c.v[2] : c.v[3];
    }
    return c;
}

// choose with whichever LDS form the n-tuple size has
template <int N>
__device__ __forceinline__ Choice choose_lds(const float* __restrict__ w, const float* hot, const Moves4& mv) {
    if constexpr (N <= 3)
        return choose_small<N, false>(w, hot, mv);       // (k_td_play: one direction at a time for n = 3)
    else
        return choose_hot<N>(w, hot, mv);
}

template <int N>
__global__ __launch_bounds__(WG) void k_eval_select(const uint4* boards, uint32_t B, const float* __restrict__ w, float* value,
                                                    uint8_t* action, float4* values4) {
    uint32_t i = blockIdx.x * WG + threadIdx.x;
    if (i >= B) return;
    Moves4 mv = all_moves(ld_board(boards, i));
    Choice c = choose<N>(w, mv);
    value[i] = c.action < 0 ? 0.0f : c.value;
    action[i] = c.action < 0 ? (uint8_t)255 : (uint8_t)c.action;
    if (values4) values4[i] = make_float4(c.v[0], c.v[1], c.v[2], c.v[3]);
}

// k_eval_select with the table's LDS copy (n = 2, 3; BASELINE config 3 is the n = 3 case): a persistent grid, one workgroup
// per CU, every thread takes boards i, i + threads, ...
#define G2048_EVAL_LDS_KERNEL(NAME, NN, ATTR)                                                                                     \
    __global__ __launch_bounds__(PLAY_HOT_WG) ATTR void NAME(const uint4* boards, uint32_t B, const float* __restrict__ w, float* value, uint8_t* action, \
                                                             float4* values4) {                                                   \
        extern __shared__ __attribute__((aligned(16))) float hot[];                                                               \
        load_hot_set<NN, PLAY_HOT_WG>(hot, w);                                                                                    \
        for (uint32_t i = blockIdx.x * PLAY_HOT_WG + threadIdx.x; i < B; i += gridDim.x * PLAY_HOT_WG) {                          \
            Moves4 mv = all_moves(ld_board(boards, i));                                                                           \
            Choice c = choose_small<NN, true>(w, hot, mv);                                                                           \
            value[i] = c.action < 0 ? 0.0f : c.value;                                                                             \
            action[i] = c.action < 0 ? (uint8_t)255 : (uint8_t)c.action;                                                          \
            if (values4) values4[i] = make_float4(c.v[0], c.v[1], c.v[2], c.v[3]);                                                \
        }                                                                                                                         \
    }
G2048_EVAL_LDS_KERNEL(k_eval_select_lds2, 2, )
G2048_EVAL_LDS_KERNEL(k_eval_select_lds3, 3, __attribute__((amdgpu_waves_per_eu(PLAY_SMALL3_WAVES_PER_EU, PLAY_SMALL3_WAVES_PER_EU))))
#undef G2048_EVAL_LDS_KERNEL

template <int N>
__device__ __forceinline__ void store_orbit_indices(uint8_t* base, uint32_t B, uint32_t i, const Packed& p) {
    static_assert(N >= 4, "orbits exist for n >= 4");
    constexpr int F = Shape<N>::F;
    uint32_t idx[6][4] = {};
#pragma unroll
    for (uint32_t g = 0; g < 8; ++g) {
        uint32_t s[F];
        feature_slots<N>(d4_image(p, g), s);            // g is constant after unrolling; what no orbit visits is dead code
#pragma unroll
        for (int v = 0; v < (N == 4 ? 5 : 6); ++v)
            if ((COSET_MASK[v] >> g) & 1u) {
                const uint32_t rel = s[ORBIT_REPS[v]] - feature_offset(N, ORBIT_REPS[v]);
                idx[v][coset_rank(COSET_MASK[v], g)] = v == 5 ? cross_order(rel) : rel;        // (the cross orbit's table: features.hpp)
            }
    }
    const OrbitIdx o = orbit_idx(base, B);
#pragma unroll
    for (int v = 0; v < 4; ++v) o.q[(size_t)v * B + i] = make_uint2(idx[v][0] | idx[v][1] << 16, idx[v][2] | idx[v][3] << 16);
    o.c[i] = (uint16_t)idx[4][0];
    if (N >= 5) o.x[i] = make_uint4(idx[5][0], idx[5][1], idx[5][2], idx[5][3]);
}

__device__ __forceinline__ void push_terminal(const TdRecs& r, const Packed& state, float dw) {
    uint32_t slot = atomicAdd(r.qcount, 1u);        // the compiler folds the wave's increments into one atomic
    st_packed(r.qstate, slot, state);
    r.qdw[slot] = dw;
}

// Step part 1 — the body of `while not game.game_over` in QAgent.episode (r_learning.py:228-246) for every live
// lane, all reading the same table.  `prev` is double-buffered: prev_cur holds `state`, prev_nxt receives this
// step's afterstate, so the main record needs no copy.
// `perm` (null on ordinary steps, when in == out): position i of the new lane order takes the lane now at position perm[i];
// the step's records stay in the OLD order (dw1 is written at perm[i], where the update kernels find the lane's `state`).
template <int N, int TPB, bool HOT, bool PERM>
__device__ __forceinline__ void td_play_body(LaneSet in, LaneSet out, const uint32_t* __restrict__ perm_, uint4* prev_nxt,
                                             uint32_t B, const float* __restrict__ w, float alpha, TdRecs recs, int auto_reset,
                                             Stats* stats, GameLog lg, uint32_t static_rounds) {
    constexpr float F = (float)Shape<N>::F;
    const uint32_t* const perm = PERM ? perm_ : nullptr;       // (compile-time: a run-time test would put a wait behind every block's first load)
    __shared__ WgStats ws;
    // (DYNAMIC shared memory: with the 100+ KB known at compile time the compiler derives "one workgroup per CU" itself and
    // drops the register budget that amdgpu_waves_per_eu asks for — see choose_hot)
    extern __shared__ __attribute__((aligned(16))) float hot[];
    wg_stats_init(&ws);
    if constexpr (HOT) load_hot_set<N, TPB>(hot, w);
    if (blockIdx.x == 0 && threadIdx.x < PLAY_SEGS) {
        recs.blocks_next[threadIdx.x * PLAY_SEG_STRIDE] = 0;
        if (threadIdx.x == 0) {
            *recs.qcount_next = 0;
            *recs.dwmax_next = 0;
        }
    }
    // Persistent WAVES: the grid is what the chip holds at once (play_grid) and every wave takes 64-lane blocks until none
    // is left — no barrier anywhere in the loop, the waves of a workgroup run on independently.  The first `static_rounds`
    // blocks of wave v are v, v + V, v + 2V ... (V waves in the grid); the blocks behind them are cut into 8 segments, one
    // per XCD (blockIdx % 8), each handed out through its own counter, which keeps the hardware's balance at the end of the
    // launch (a wave that is ahead takes more) without thousands of returning atomics on one address.
    // The loop is software-pipelined by one block: while block k is computed, the lane state of block k + 1 is already on
    // its way (two waves per SIMD do not hide a load round trip in front of every block), so the block index is needed one
    // block early and the dynamic counter is read TWO blocks ahead (vmcnt retires in order: an atomic issued behind a
    // block's loads and stores would wait for all of them).
    constexpr uint32_t WAVES = TPB / 64, NONE = 0xFFFFFFFFu;
    const uint32_t lane = threadIdx.x & 63u;
    // (wave-uniform by construction; readfirstlane tells the compiler, so that block numbers live in SGPRs and the loop's
    // exits are scalar branches)
    const uint32_t nwaves = gridDim.x * WAVES, wave_id = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * WAVES + (threadIdx.x >> 6)));
    const uint32_t nblocks = (B + 63u) / 64u;
    const uint32_t dyn0 = static_rounds * nwaves < nblocks ? static_rounds * nwaves : nblocks;     // first block of the dynamic region
    const uint32_t nseg = (gridDim.x + 7u) / 8u < PLAY_SEGS ? (gridDim.x + 7u) / 8u : PLAY_SEGS;
    const uint32_t seg = (blockIdx.x / 8u) % nseg, seg_len = (nblocks - dyn0 + nseg - 1u) / nseg;
    const uint32_t seg_lo = dyn0 + seg * seg_len, seg_hi = seg_lo + seg_len < nblocks ? seg_lo + seg_len : nblocks;
    // The counter's address is made opaque to the compiler: for an atomic on a uniform address it would elect a lane and
    // broadcast the result with v_readfirstlane behind an s_waitcnt vmcnt(0) right at the atomic — a full round trip, plus
    // the drain of the previous block's stores, in front of every block (which is what the dynamic rounds used to cost).
    uint32_t opaque_zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(opaque_zero));
    uint32_t* const counter = recs.blocks + seg * PLAY_SEG_STRIDE + opaque_zero;
    uint32_t ahead = 0, my_moves = 0, my_dirs = 0;
    float dw_big = 0.0f;            // largest |dw| this lane emits in the whole launch

    // the state a lane carries into a block
    struct LaneIn {
        uint32_t src;
        uint8_t fl;
        Board b;
        Rng g;
        int32_t score;
        float old_label;
        uint32_t lid;
    };
    // the whole lane state is requested at once: loading the flags first and the rest behind the DONE test would put two
    // memory round trips in front of every block
    // (unconditional loads from a clamped index: loads under a branch are merged with the "not loaded" value at its end,
    // which makes the compiler wait for them right there)
    auto load_lane = [&](uint32_t blk) {
        LaneIn L;
        const uint32_t i = blk * 64u + lane, j = (blk != NONE && i < B) ? i : 0u;
        L.src = perm ? perm[j] : j;
        L.fl = in.flags[L.src];
        L.b = ld_board(in.boards, L.src);
        L.g = ld_rng(in.rng, L.src);
        L.score = in.scores[L.src];
        L.old_label = in.label[L.src];
        const uint32_t id = in.lane_id[L.src];          // (always loaded: a load under a run-time test is waited for at the test's end)
        L.lid = (PERM || lg.lanes) ? id : i;
        return L;
    };
    uint32_t blk;                   // the block being computed
    if (static_rounds > 0) {
        blk = wave_id;
    } else {
        if (lane == 0) ahead = atomicAdd(counter, 1u);
        blk = seg_lo + (uint32_t)__builtin_amdgcn_readfirstlane((int)ahead);
        if (blk >= seg_hi) blk = NONE;
    }
    if (static_rounds <= 1 && lane == 0) ahead = atomicAdd(counter, 1u);        // the second block comes from the counter
    // the block behind block number `it` of this wave (and, two blocks ahead, the request to the counter)
    auto block_after = [&](uint32_t it) {
        uint32_t nxt;
        if (it + 1 < static_rounds) {
            nxt = (it + 1) * nwaves + wave_id;
        } else {
            nxt = seg_lo + (uint32_t)__builtin_amdgcn_readfirstlane((int)ahead);      // (seg_lo is added here, not at the atomic: an
            if (nxt >= seg_hi) nxt = NONE;                                              // instruction that consumes its result makes the wave wait for it there)
        }
        if (it + 2 >= static_rounds && lane == 0) ahead = atomicAdd(counter, 1u);
        return nxt;
    };
    auto process = [&](const uint32_t blk, const LaneIn& cur) __attribute__((always_inline)) {
    // every loaded register is "used" here on all paths: a wave whose lanes skip the block (past the batch, finished games)
    // would otherwise carry unwaited loads to the loop head, where the compiler then drains the whole memory queue — the
    // block's own stores included — before it reuses their registers
    asm volatile("" ::"v"((uint32_t)cur.fl), "v"(cur.b.r[0]), "v"(cur.b.r[1]), "v"(cur.b.r[2]), "v"(cur.b.r[3]), "v"(cur.g.s0), "v"(cur.g.s1), "v"(cur.score),
                 "v"(cur.old_label), "v"(cur.lid));
    const uint32_t i = blk * 64u + lane;

    bool moved = false;
    uint32_t ndirs = 0;             // directions that were open to this lane's move
    if (i < B) {
        const uint32_t src = cur.src;
        uint8_t fl = cur.fl;
        Board b = cur.b;
        Rng g = cur.g;
        int32_t score = cur.score;
        float old_label = cur.old_label;
        const uint32_t lid = cur.lid;
        if (perm) out.lane_id[i] = lid;
        float dw1 = 0.0f;
        uint32_t lm = 0;        // what this lane did: bits 0-1 direction, 2 moved, 4-7 new tile's cell, 8-9 new tile, 10 spawned, 11 game ended
        if (fl & DONE) {
            if (perm) {         // a finished lane moves with the others
                st_board(out.boards, i, b);
                st_rng(out.rng, i, g);
                out.scores[i] = score;
                out.label[i] = old_label;
                out.flags[i] = fl;
                prev_nxt[i] = recs.state1[src];
            }
        } else {
            Moves4 mv = all_moves(b);
            Choice c;
            if constexpr (HOT)
                c = choose_lds<N>(w, hot, mv);
            else
                c = choose<N>(w, mv);
            bool over, overflow = false;
            if (c.action >= 0) {
                Moved ch = pick(mv, (uint32_t)c.action);
                int32_t reward = (int32_t)merged_score(ch.ma, ch.mb);
                if (fl & HAS_PREV) dw1 = ((float)reward + c.value - old_label) * alpha / F;
                score += reward;
                Packed after = pack_board(ch.after);
                st_packed(prev_nxt, i, after);
                if constexpr (N >= 4) store_orbit_indices<N>(recs.oidx_nxt, B, i, after);
                old_label = c.value;
                fl |= HAS_PREV;
                moved = true;
                ndirs = popcount32(changed_mask(mv));
                b = ch.after;
                lm = (uint32_t)c.action | 4u;
                if (spawn(b, g)) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const uint32_t d = b.r[r] ^ ch.after.r[r];          // the one byte that changed
                        if (d) {
                            const uint32_t col = (uint32_t)(__ffs((int)d) - 1) >> 3;
                            lm |= ((uint32_t)(4 * r) + col) << 4 | ((d >> (8 * col)) & 3u) << 8 | 1u << 10;
                        }
                    }
                }
                overflow = max_tile(b) >= 16u;
                over = game_over(b) || overflow;
                if (over) {
                    const float dw2 = -c.value * alpha / F;
                    if (isfinite(dw2)) {
                        push_terminal(recs, after, dw2);
                        dw_big = fmaxf(dw_big, fabsf(dw2));
                    } else {
                        atomicAdd(&ws.nonfinite, 1u);
                    }
                }
            } else {
                // a dead board was loaded: the reference's loop would not run; only the terminal update remains
                over = true;
                prev_nxt[i] = recs.state1[src];
                if (fl & HAS_PREV) {
                    const float dw2 = -old_label * alpha / F;
                    if (isfinite(dw2)) {
                        push_terminal(recs, ld_packed(recs.state1, src), dw2);
                        dw_big = fmaxf(dw_big, fabsf(dw2));
                    } else {
                        atomicAdd(&ws.nonfinite, 1u);
                    }
                }
            }
            const int32_t final_score = score;
            const Board final_board = b;
            if (over) {
                lm |= 1u << 11;
                count_finished(&ws, b, score, overflow);
                if (auto_reset) {
                    b = new_game(g);
                    score = 0;
                    old_label = 0.0f;
                    fl &= (uint8_t)~HAS_PREV;
                } else {
                    fl |= DONE;
                }
            }
            if (lid < lg.lanes) log_step(lg, lid, lm, moved, over, final_score, over && auto_reset, b, final_board);
            st_board(out.boards, i, b);
            st_rng(out.rng, i, g);
            G2048_ST(out.scores + i, score);
            G2048_ST(out.label + i, old_label);
            G2048_ST(out.flags + i, fl);
        }
        // a record whose dw is not finite (a table poisoned with inf / NaN) is dropped and counted: the fixed-point sums
        // of the LDS-owner update have no encoding for it
        if (!isfinite(dw1)) {
            atomicAdd(&ws.nonfinite, 1u);
            dw1 = 0.0f;
        }
        recs.dw1[src] = dw1;
        G2048_ST(out.last_move + i, (uint16_t)lm);
        dw_big = fmaxf(dw_big, fabsf(dw1));
    }
    my_moves += moved ? 1u : 0u;
    my_dirs += ndirs;
    };
    // two copies of the body with the roles of the two state sets swapped: with one copy the prefetched state would have to
    // be moved into the current one's registers at the end of every block, behind a wait for the block's own stores
    if (blk != NONE) {
        LaneIn sa = load_lane(blk), sb;
        for (uint32_t it = 0;; it += 2) {
            const uint32_t nb = block_after(it);
            sb = load_lane(nb);
            process(blk, sa);
            if (nb == NONE) break;
            blk = block_after(it + 1);
            sa = load_lane(blk);
            process(nb, sb);
            if (blk == NONE) break;
        }
    }
    // the wave reductions happen once per launch, not once per block (they cost 2.4 us of a block's 22 there)
    if (dw_big > 0.0f) atomicMax(&ws.dw_max_bits, __float_as_uint(dw_big));
    count_moves(&ws, my_moves, my_dirs);
    wg_stats_flush(&ws, stats);
    if (threadIdx.x == 0 && ws.dw_max_bits) atomicMax(recs.dwmax, ws.dw_max_bits);
}

// The two kernels around the body.  The hot-set form is compiled with a capped register allocator: the VGPRs above the cap
// hold its gathers' results and are named only in choose_hot's asm statements (see there).
template <int N, int TPB, bool PERM>
__global__ __launch_bounds__(TPB) void k_td_play(LaneSet in, LaneSet out, const uint32_t* __restrict__ perm, uint4* prev_nxt, uint32_t B,
                                                                       const float* __restrict__ w, float alpha, TdRecs recs, int auto_reset, Stats* stats,
                                                                       GameLog lg, uint32_t static_rounds) {
    td_play_body<N, TPB, false, PERM>(in, out, perm, prev_nxt, B, w, alpha, recs, auto_reset, stats, lg, static_rounds);
}
// (n = 2: LDS reads only, no registers outside the allocator's range; n = 3: see choose_small)
template <int N, int TPB, bool PERM>
__global__ __launch_bounds__(TPB) void k_td_play_lds2(LaneSet in, LaneSet out, const uint32_t* __restrict__ perm, uint4* prev_nxt, uint32_t B,
                                                                            const float* __restrict__ w, float alpha, TdRecs recs, int auto_reset, Stats* stats,
                                                                            GameLog lg, uint32_t static_rounds) {
    static_assert(N == 2, "n = 2");
    td_play_body<N, TPB, true, PERM>(in, out, perm, prev_nxt, B, w, alpha, recs, auto_reset, stats, lg, static_rounds);
}
template <int N, int TPB, bool PERM>
__global__ __launch_bounds__(TPB) __attribute__((amdgpu_waves_per_eu(PLAY_SMALL3P_WAVES_PER_EU, PLAY_SMALL3P_WAVES_PER_EU))) void k_td_play_lds3(
    LaneSet in, LaneSet out, const uint32_t* __restrict__ perm, uint4* prev_nxt, uint32_t B, const float* __restrict__ w, float alpha, TdRecs recs,
    int auto_reset, Stats* stats, GameLog lg, uint32_t static_rounds) {
    static_assert(N == 3, "n = 3");
    td_play_body<N, TPB, true, PERM>(in, out, perm, prev_nxt, B, w, alpha, recs, auto_reset, stats, lg, static_rounds);
}
template <int N, int TPB, bool PERM>
__global__ __launch_bounds__(TPB) __attribute__((amdgpu_waves_per_eu(PLAY_HOT_WAVES_PER_EU, PLAY_HOT_WAVES_PER_EU))) void k_td_play_hot(
    LaneSet in, LaneSet out, const uint32_t* __restrict__ perm, uint4* prev_nxt, uint32_t B, const float* __restrict__ w, float alpha, TdRecs recs,
    int auto_reset, Stats* stats, GameLog lg, uint32_t static_rounds) {
    td_play_body<N, TPB, true, PERM>(in, out, perm, prev_nxt, B, w, alpha, recs, auto_reset, stats, lg, static_rounds);
}

// a kernel that takes more than 64 KB of dynamic LDS has to be told so once (per process: the attribute belongs to the function)
// (keyed on the kernel's address and the device — every k_td_play_* instantiation has the same function TYPE, so a
// per-type flag would cover only the first of them: the advisor's round-3 finding)
template <class K>
void allow_dynamic_lds(K kernel, uint32_t bytes) {
    if (bytes <= 65536u) return;
    static std::mutex mu;
    static std::vector<std::pair<const void*, int>> done;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const std::pair<const void*, int> key{reinterpret_cast<const void*>(kernel), dev};
    std::lock_guard<std::mutex> lock(mu);
    if (std::find(done.begin(), done.end(), key) != done.end()) return;
    const hipError_t e = hipFuncSetAttribute(key.first, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) fprintf(stderr, "[g2048] hipFuncSetAttribute(MaxDynamicSharedMemorySize = %u): %s\n", bytes, hipGetErrorString(e));
    done.push_back(key);
}

}  // namespace

namespace g2048 {

int play_blocks_per_cu(int n, bool hot) {
    int per_cu = 0;
    hipError_t e;
    if (hot && n == 2)
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_td_play_lds2<2, PLAY_HOT_WG, false>, PLAY_HOT_WG, HotShape<2>::SLOTS * 4u);
    else if (hot && n == 3)
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_td_play_lds3<3, PLAY_HOT_WG, false>, PLAY_HOT_WG, HotShape<3>::SLOTS * 4u);
    else if (hot && n == 4)
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_td_play_hot<4, PLAY_HOT_WG, false>, PLAY_HOT_WG, HotShape<4>::SLOTS * 4u);
    else if (hot)
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_td_play_hot<5, PLAY_HOT_WG, false>, PLAY_HOT_WG, HotShape<5>::SLOTS * 4u);
    else
        switch (n) {
            case 2: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_td_play<2, PLAY_WG, false>, PLAY_WG, 0); break;
            case 3: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_td_play<3, PLAY_WG, false>, PLAY_WG, 0); break;
            case 4: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_td_play<4, PLAY_WG, false>, PLAY_WG, 0); break;
            case 5: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_td_play<5, PLAY_WG, false>, PLAY_WG, 0); break;
            default: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_td_play<6, PLAY_WG, false>, PLAY_WG, 0); break;
        }
    return e == hipSuccess && per_cu > 0 ? per_cu : 0;
}

hipError_t launch_td_play(hipStream_t st, int n, bool hot, unsigned grid, LaneSet lin, LaneSet lout, const uint32_t* perm, uint4* pn, uint32_t B, const float* w,
                          float alpha, const TdRecs& recs, int auto_reset, Stats* stats, const GameLog& lg, uint32_t static_rounds) {
#define G2048_PLAY_(KERNEL, NN, TPB, HOT, PERM)                                                                                      \
    (allow_dynamic_lds(KERNEL<NN, TPB, PERM>, HOT ? HotShape<NN>::SLOTS * 4u : 0u),                                                  \
     KERNEL<NN, TPB, PERM><<<grid, TPB, HOT ? HotShape<NN>::SLOTS * 4u : 0u, st>>>(lin, lout, perm, pn, B, w, alpha, recs, auto_reset, stats, lg, static_rounds))
#define G2048_PLAY(KERNEL, NN, TPB, HOT) (perm ? (G2048_PLAY_(KERNEL, NN, TPB, HOT, true)) : (G2048_PLAY_(KERNEL, NN, TPB, HOT, false)))
    if (hot) {
        switch (n) {
            case 2: G2048_PLAY(k_td_play_lds2, 2, PLAY_HOT_WG, true); break;
            case 3: G2048_PLAY(k_td_play_lds3, 3, PLAY_HOT_WG, true); break;
            case 4: G2048_PLAY(k_td_play_hot, 4, PLAY_HOT_WG, true); break;
            case 5: G2048_PLAY(k_td_play_hot, 5, PLAY_HOT_WG, true); break;
            default: return hipErrorInvalidValue;
        }
    } else {
        switch (n) {
            case 2: G2048_PLAY(k_td_play, 2, PLAY_WG, false); break;
            case 3: G2048_PLAY(k_td_play, 3, PLAY_WG, false); break;
            case 4: G2048_PLAY(k_td_play, 4, PLAY_WG, false); break;
            case 5: G2048_PLAY(k_td_play, 5, PLAY_WG, false); break;
            case 6: G2048_PLAY(k_td_play, 6, PLAY_WG, false); break;
            default: return hipErrorInvalidValue;
        }
    }
#undef G2048_PLAY
#undef G2048_PLAY_
    return hipGetLastError();
}

hipError_t launch_eval_select(hipStream_t st, int n, bool lds, unsigned grid, const uint4* boards, uint32_t B, const float* w, float* value, uint8_t* action,
                              float4* values4) {
    if (lds && n == 2) {
        k_eval_select_lds2<<<grid, PLAY_HOT_WG, HotShape<2>::SLOTS * 4u, st>>>(boards, B, w, value, action, values4);
    } else if (lds && n == 3) {
        allow_dynamic_lds(k_eval_select_lds3, HotShape<3>::SLOTS * 4u);
        k_eval_select_lds3<<<grid, PLAY_HOT_WG, HotShape<3>::SLOTS * 4u, st>>>(boards, B, w, value, action, values4);
    } else {
        const unsigned g = (unsigned)(((uint64_t)B + WG - 1) / WG);
        switch (n) {
            case 2: k_eval_select<2><<<g, WG, 0, st>>>(boards, B, w, value, action, values4); break;
            case 3: k_eval_select<3><<<g, WG, 0, st>>>(boards, B, w, value, action, values4); break;
            case 4: k_eval_select<4><<<g, WG, 0, st>>>(boards, B, w, value, action, values4); break;
            case 5: k_eval_select<5><<<g, WG, 0, st>>>(boards, B, w, value, action, values4); break;
            case 6: k_eval_select<6><<<g, WG, 0, st>>>(boards, B, w, value, action, values4); break;
            default: return hipErrorInvalidValue;
        }
    }
    return hipGetLastError();
}

}  // namespace g2048
