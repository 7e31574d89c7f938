// Board arithmetic for the batched 2048 environment — pure integer C++, usable from HIP device code
// (and, for tests/hostcheck only, from a plain host compile: there is no CPU product path).
//
// Layout: a board is 16 uint8 log2-tiles, row-major (game2048/game_logic.py:62 uses int32[4,4]); in
// registers it is four 32-bit row words, byte j of word r = cell (r, j).  One lane owns one board.
//
// Moves are SWAR: the four lines that a direction slides are handled at once, one byte lane per line.
// P0..P3 are the four "position words": P0 holds, for each of the four lines, the cell nearest to the
// wall the tiles travel to.  For up/down the position words are the row words themselves; for
// left/right they are the column words (4x4 byte transpose).  Semantics restated from
// create_table (game_logic.py:18-39): pack non-zeros toward the wall, ONE pass from the wall merging
// equal neighbours x,x -> x+1 (score += 2^(x+1)), pack again; a merged tile never merges twice.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define G2048_HD __host__ __device__ __forceinline__
#else
#define G2048_HD inline
#endif

namespace g2048 {

struct Board {
    uint32_t r[4];
};

// ---- SWAR byte helpers (all tile bytes are < 0x80)

// 0x80 in every byte lane where x's byte is non-zero (no cross-byte carry: bytes are < 0x80)
G2048_HD uint32_t nonzero_hi(uint32_t x) { return (x + 0x7F7F7F7Fu) & 0x80808080u; }

// 0xFF in every byte lane where x's byte is zero
G2048_HD uint32_t zero_mask(uint32_t x) {
    const uint32_t h = nonzero_hi(x);
    return ~(h | (h - (h >> 7)));           // (no multiply: v_mul_lo_u32 issues at a quarter of the rate)
}

// 4x4 byte transpose: column words from row words (an involution)
G2048_HD void transpose(const uint32_t in[4], uint32_t out[4]) {
    uint32_t a = in[0], b = in[1], c = in[2], d = in[3];
#if defined(__HIP_DEVICE_COMPILE__)
    // eight byte permutes (v_perm_b32: selector byte k picks byte k of {second operand, first operand})
    const uint32_t t0 = __builtin_amdgcn_perm(b, a, 0x05010400u), t1 = __builtin_amdgcn_perm(b, a, 0x07030602u);     // a0 b0 a1 b1 | a2 b2 a3 b3
    const uint32_t t2 = __builtin_amdgcn_perm(d, c, 0x05010400u), t3 = __builtin_amdgcn_perm(d, c, 0x07030602u);     // c0 d0 c1 d1 | c2 d2 c3 d3
    out[0] = __builtin_amdgcn_perm(t2, t0, 0x05040100u);
    out[1] = __builtin_amdgcn_perm(t2, t0, 0x07060302u);
    out[2] = __builtin_amdgcn_perm(t3, t1, 0x05040100u);
    out[3] = __builtin_amdgcn_perm(t3, t1, 0x07060302u);
#else
    out[0] = (a & 0xFFu) | ((b & 0xFFu) << 8) | ((c & 0xFFu) << 16) | (d << 24);
    out[1] = ((a >> 8) & 0xFFu) | (b & 0xFF00u) | ((c & 0xFF00u) << 8) | ((d & 0xFF00u) << 16);
    out[2] = ((a >> 16) & 0xFFu) | ((b >> 8) & 0xFF00u) | (c & 0xFF0000u) | ((d & 0xFF0000u) << 8);
    out[3] = (a >> 24) | ((b >> 16) & 0xFF00u) | ((c >> 8) & 0xFF0000u) | (d & 0xFF000000u);
#endif
}

// if a's byte is empty take b's byte (b's becomes empty) — per byte lane.  The masks are 7 bits wide (0x7F where a holds a
// tile): enough for bytes below 0x80, and one subtraction instead of a widening multiply.
G2048_HD void pull(uint32_t& a, uint32_t& b) {
    const uint32_t h = nonzero_hi(a), keep = h - (h >> 7);
    a |= b & ~keep;
    b &= keep;
}

// one merge step: where the byte lanes of a and b hold the same tile, a's tile grows by one and b's lane empties;
// returns 0x7F in those lanes.  (The "a != 0" guard stops padding zeros, and a just-emptied cell, from merging.)
G2048_HD uint32_t merge(uint32_t& a, uint32_t& b) {
    const uint32_t e = ~nonzero_hi(a ^ b) & nonzero_hi(a), one = e >> 7, e7 = e - one;
    a += one;
    b &= ~e7;
    return e7;
}

// Slide the four lines toward position 0.  Returns the new position words and the merged-tile values.
G2048_HD void slide(uint32_t p0, uint32_t p1, uint32_t p2, uint32_t p3, uint32_t out[4], uint32_t& merged_a,
                    uint32_t& merged_b) {
    // pack toward position 0 (bubble network: 6 conditional pulls)
    pull(p0, p1); pull(p1, p2); pull(p2, p3);
    pull(p0, p1); pull(p1, p2);
    pull(p0, p1);
    // one merge pass from the wall
    const uint32_t e0 = merge(p0, p1);
    const uint32_t e1 = merge(p1, p2);
    const uint32_t e2 = merge(p2, p3);
    // per byte lane a merge at 0 excludes one at 1, and one at 1 excludes one at 2: two words hold them all
    merged_a = (p0 & e0) | (p1 & e1);
    merged_b = (p2 & e2);
    // close the gaps the merges opened (only positions 1 and 2 can be gaps in front of a tile)
    pull(p1, p2); pull(p2, p3);
    out[0] = p0; out[1] = p1; out[2] = p2; out[3] = p3;
}

// score of a move = sum of 2^v over merged tiles v (game_logic.py:31-33)
G2048_HD uint32_t merged_score(uint32_t ma, uint32_t mb) {
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        uint32_t va = (ma >> (8 * j)) & 0xFFu, vb = (mb >> (8 * j)) & 0xFFu;
        s += (1u << (va & 31u)) & ~1u;      // v = 0 (no merge) contributes 0; merged tiles are >= 2
        s += (1u << (vb & 31u)) & ~1u;
    }
    return s;
}

struct Moved {
    Board after;
    uint32_t ma, mb;    // merged-tile words (score = merged_score(ma, mb))
    bool changed;
};

// One direction of Game.pre_move (game_logic.py:136-142): 0 left, 1 up, 2 right, 3 down (Game.actions, :50).
// `rows` are the row words, `cols` the column words (transpose of rows).
template <int DIR>
G2048_HD Moved move_dir(const uint32_t rows[4], const uint32_t cols[4]) {
    Moved m;
    uint32_t out[4];
    if (DIR == 0) {                 // left: lines are rows, wall at column 0 -> position words are columns 0..3
        slide(cols[0], cols[1], cols[2], cols[3], out, m.ma, m.mb);
        uint32_t c[4] = {out[0], out[1], out[2], out[3]};
        transpose(c, m.after.r);
    } else if (DIR == 2) {          // right: wall at column 3
        slide(cols[3], cols[2], cols[1], cols[0], out, m.ma, m.mb);
        uint32_t c[4] = {out[3], out[2], out[1], out[0]};
        transpose(c, m.after.r);
    } else if (DIR == 1) {          // up: lines are columns, wall at row 0 -> position words are rows 0..3
        slide(rows[0], rows[1], rows[2], rows[3], out, m.ma, m.mb);
        m.after.r[0] = out[0]; m.after.r[1] = out[1]; m.after.r[2] = out[2]; m.after.r[3] = out[3];
    } else {                        // down: wall at row 3
        slide(rows[3], rows[2], rows[1], rows[0], out, m.ma, m.mb);
        m.after.r[0] = out[3]; m.after.r[1] = out[2]; m.after.r[2] = out[1]; m.after.r[3] = out[0];
    }
    m.changed = ((m.after.r[0] ^ rows[0]) | (m.after.r[1] ^ rows[1]) | (m.after.r[2] ^ rows[2]) |
                 (m.after.r[3] ^ rows[3])) != 0;
    return m;
}

// ---- terminal test and spawn (game_logic.py:96-121)

G2048_HD uint32_t popcount32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __popc(x);
#else
    return (uint32_t)__builtin_popcount(x);
#endif
}

// 16-bit mask, bit (4r + c) set iff cell (r, c) is empty — row-major, as Game.empty lists them (:96-99)
G2048_HD uint32_t empty_bits(const Board& b) {
    uint32_t bits = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        uint32_t z = zero_mask(b.r[r]) & 0x80808080u;               // bit 7 of each empty byte
        // gather bits 7,15,23,31 into a nibble: multiply trick
        uint32_t nib = ((z >> 7) * 0x10204080u) >> 28;              // b0->bit0, b1->bit1, b2->bit2, b3->bit3
        bits |= nib << (4 * r);
    }
    return bits;
}

G2048_HD uint32_t empty_count(const Board& b) { return popcount32(empty_bits(b)); }

// number of equal neighbour pairs, horizontal + vertical (adjacent_pair_count, game_logic.py:105-107)
G2048_HD uint32_t adjacent_pairs(const Board& b) {
    uint32_t n = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        uint32_t x = b.r[r] ^ (b.r[r] >> 8);                        // bytes 0..2: cell c vs c+1
        n += popcount32(zero_mask(x) & 0x00808080u);                // byte 3 (cell 3 vs nothing) ignored
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) n += popcount32(zero_mask(b.r[r] ^ b.r[r + 1]) & 0x80808080u);
    return n;
}

G2048_HD bool game_over(const Board& b) { return empty_bits(b) == 0 && adjacent_pairs(b) == 0; }

G2048_HD uint32_t max_tile(const Board& b) {
    uint32_t m = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            uint32_t v = (b.r[r] >> (8 * c)) & 0xFFu;
            m = v > m ? v : m;
        }
    return m;
}

// position (0..15) of the k-th set bit of a 16-bit mask, k < popcount(mask)
G2048_HD uint32_t kth_set_bit(uint32_t mask, uint32_t k) {
    uint32_t pos = 0;
    // binary search over halves by popcount
    uint32_t c = popcount32(mask & 0xFFu);
    if (k >= c) { k -= c; pos += 8; mask >>= 8; }
    c = popcount32(mask & 0xFu);
    if (k >= c) { k -= c; pos += 4; mask >>= 4; }
    c = popcount32(mask & 0x3u);
    if (k >= c) { k -= c; pos += 2; mask >>= 2; }
    c = mask & 1u;
    if (k >= c) { pos += 1; }
    return pos;
}

// new_tile with injected draws (game_logic.py:112-121): tile 2 iff r10 == 0, at the k-th empty cell (row-major)
G2048_HD void place_tile(Board& b, uint32_t r10, uint32_t k, uint32_t empties) {
    uint32_t pos = kth_set_bit(empties, k);
    uint32_t tile = r10 == 0 ? 2u : 1u;
    uint32_t row = pos >> 2, col = pos & 3u;
    uint32_t add = tile << (8 * col);
#pragma unroll
    for (int r = 0; r < 4; ++r) b.r[r] |= (row == (uint32_t)r) ? add : 0u;
}

// ---- per-lane RNG: xoroshiro128++ (spec mirrored in 2048_amd/rng.py)

struct Rng {
    uint64_t s0, s1;
};

G2048_HD uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

G2048_HD uint64_t splitmix64(uint64_t& x) {
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

G2048_HD Rng seed_lane(uint64_t seed, uint64_t global_lane) {
    uint64_t x = seed + global_lane;
    Rng g;
    g.s0 = splitmix64(x);
    g.s1 = splitmix64(x);
    if ((g.s0 | g.s1) == 0) g.s0 = 1;
    return g;
}

G2048_HD uint64_t next_u64(Rng& g) {
    uint64_t s0 = g.s0, s1 = g.s1;
    uint64_t u = rotl64(s0 + s1, 17) + s0;
    s1 ^= s0;
    g.s0 = rotl64(s0, 49) ^ s1 ^ (s1 << 21);
    g.s1 = rotl64(s1, 28);
    return u;
}

// (r10, k) from one draw: Lemire multiply-high on the two 32-bit halves
G2048_HD void spawn_draw(uint64_t u, uint32_t n_empty, uint32_t& r10, uint32_t& k) {
    r10 = (uint32_t)(((u >> 32) * 10ull) >> 32);
    k = (uint32_t)(((u & 0xFFFFFFFFull) * (uint64_t)n_empty) >> 32);
}

G2048_HD uint32_t pick_draw(uint64_t u, uint32_t n_valid) { return (uint32_t)(((u >> 32) * (uint64_t)n_valid) >> 32); }

// spawn from the lane's own stream; returns false (and leaves the stream untouched) if the board is full
G2048_HD bool spawn(Board& b, Rng& g, uint32_t* out_r10 = nullptr, uint32_t* out_k = nullptr) {
    uint32_t e = empty_bits(b);
    if (e == 0) return false;
    uint32_t r10, k;
    spawn_draw(next_u64(g), popcount32(e), r10, k);
    place_tile(b, r10, k, e);
    if (out_r10) *out_r10 = r10;
    if (out_k) *out_k = k;
    return true;
}

// Game.__init__ (game_logic.py:55-66): empty board + two spawns
G2048_HD Board new_game(Rng& g) {
    Board b = {{0u, 0u, 0u, 0u}};
    spawn(b, g);
    spawn(b, g);
    return b;
}

}  // namespace g2048
