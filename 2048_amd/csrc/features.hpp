// n-tuple index encoders f_2 .. f_6 (game2048/r_learning.py:17-69) and the D4 images of QAgent.update
// (r_learning.py:207-214) on a nibble-packed board.  Pure integer C++ (device code; host compile only for
// tests/hostcheck).
//
// Packed form: R[r] = x[r,0]<<12 | x[r,1]<<8 | x[r,2]<<4 | x[r,3]   (row r, leftmost cell most significant)
//              C[c] = x[0,c]<<12 | x[1,c]<<8 | x[2,c]<<4 | x[3,c]   (column c, top cell most significant)
// which are exactly the reference's "x_hor" / "x_vert" 4-tuple indices (r_learning.py:41-42), so every other
// feature is a bit-field of them.  In this form the dihedral group is nearly free: transpose swaps R and C,
// a left-right mirror reverses the nibbles of every R and the order of C, an up-down mirror the converse.
//
// Flat weight table: feature-major, groups in weight_signature order (r_learning.py:136-149):
//   n = 2,3,4: num_feat x 16^n;  n = 5: 17 x 16^4 then 4 x 16^5;  n = 6: ... then 12 x 14^6.
#pragma once
#include "board_ops.hpp"

namespace g2048 {

struct Packed {
    uint32_t R[4];
    uint32_t C[4];
};

template <int N> struct Shape;
template <> struct Shape<2> { static constexpr int F = 24; static constexpr uint32_t SLOTS = 24u * 256u; };
template <> struct Shape<3> { static constexpr int F = 52; static constexpr uint32_t SLOTS = 52u * 4096u; };
template <> struct Shape<4> { static constexpr int F = 17; static constexpr uint32_t SLOTS = 17u * 65536u; };
template <> struct Shape<5> { static constexpr int F = 21; static constexpr uint32_t SLOTS = 17u * 65536u + 4u * 1048576u; };
template <> struct Shape<6> { static constexpr int F = 33; static constexpr uint32_t SLOTS = 17u * 65536u + 4u * 1048576u + 12u * 7529536u; };

constexpr uint32_t QUAD_BASE = 0u;                              // 17 four-cell features
constexpr uint32_t CROSS_BASE = 17u * 65536u;                   // 4 five-cell features
constexpr uint32_t HEX_BASE = CROSS_BASE + 4u * 1048576u;       // 12 six-cell base-14 features
constexpr uint32_t HEX_SIZE = 7529536u;                         // 14^6

// ---- where a table entry lives in device memory (n >= 4, the four- and five-cell tables)
// A board's index into a four-cell table is four nibbles, and tiles are small: bits 2 and 3 of a nibble hardly vary.  In
// index order a 128-byte cache line holds one cell's whole nibble plus one bit of the next, so the entries a batch of boards
// reads are spread thinly over many lines.  In memory the low 16 index bits are therefore stored 4 x 4 bit-TRANSPOSED
// ([bit 3 of the four cells | bit 2 .. | bit 1 .. | bit 0 ..]: a line now holds the entries that differ in the cells' low
// bits; the map is its own inverse).  Every consumer of the table goes through table_place(); the ABI
// (g2048_weights_get / _set, the delta buffers, feature indices) stays in the reference's order.  Measured: k_td_play
// 0.168 -> 0.156 ms.  (A per-table rotation of index bits 5..10 on top of it — the tables are 256 KB
// apart, so their busy lines share cache sets — looked good in a timing-only build and costs 1-2 % in the real one:
// profiles/r02_knob_ab.txt.)
G2048_HD uint32_t bit_transpose16(uint32_t v) {       // the low 16 bits as a 4 x 4 bit matrix, transposed; upper bits unchanged
    uint32_t x = v & 0xFFFFu;
    uint32_t t = (x ^ (x >> 3)) & 0x0A0Au;
    x ^= t ^ (t << 3);
    t = (x ^ (x >> 6)) & 0x00CCu;
    x ^= t ^ (t << 6);
    return (v & ~0xFFFFu) | x;
}
G2048_HD uint32_t table_place(uint32_t slot) { return bit_transpose16(slot); }

// ---- the order of the cross orbit's accumulation table (five cells, 20 index bits): all five cells' bits transposed,
// [bit 3 of cells 0..4 | bit 2 .. | bit 1 .. | bit 0 ..].  The table is cut into chunks by its TOP bits, which are then the
// cells' high bits: nearly all of a step's adds fall into the first two of its 64 chunks instead of into ten to twenty,
// and a chunk costs a scan of all records.  (The member tables in memory keep table_place: a gather wants the low bits of
// FOUR cells inside a line, a chunk wants the high bits of ALL cells on top.)
G2048_HD uint32_t cross_order(uint32_t k) {
    const uint32_t t = bit_transpose16(k & 0xFFFFu), c4 = k >> 16;          // t: four planes of four bits; c4: the fifth cell
    uint32_t out = 0;
    for (uint32_t b = 0; b < 4u; ++b) out |= (((t >> (4u * b)) & 15u) | (((c4 >> b) & 1u) << 4)) << (5u * b);
    return out;
}
G2048_HD uint32_t cross_unorder(uint32_t o) {
    uint32_t t = 0, c4 = 0;
    for (uint32_t b = 0; b < 4u; ++b) {
        const uint32_t plane = (o >> (5u * b)) & 31u;
        t |= (plane & 15u) << (4u * b);
        c4 |= (plane >> 4) << b;
    }
    return (c4 << 16) | bit_transpose16(t);
}

// ---- the order of the FOUR-cell orbits' accumulation tables (n >= 4).  An LDS owner holds 16 384 fixed-point slots and every
// chunk that is held costs a scan of all records, so the order wants all of a step's adds in as few chunks as possible.  Cut by
// the index's top bits (the leading cell), young boards spread over 2 - 3 of an orbit's 4 chunks (0.75 - 0.83 / 0.16 - 0.23 /
// 0.01 - 0.03 of the adds).  Instead the 11^4 = 14 641 indices whose four cells are all <= 10 (tiles up to 1 024) come first, as
// base-11 numbers — chunk 0, with EVERY add of an agent that has not made a 2 048 yet and 0.70 - 0.99 of a trained one's — and
// behind them, from slot QUAD_HOT on, the whole 16-bit index space in index order, of which only the entries with a cell >= 11 are
// ever used (the others are holes: 5 chunks per orbit instead of 4, 80 KB of D more).  Records keep the plain 16-bit indices; the
// owner kernel maps them, two indices per 32-bit word at a time (own_accum_quad).
constexpr uint32_t QUAD_HOT = 16384u, QUAD_DSIZE = QUAD_HOT + 65536u, QUAD_HOT_USED = 14641u;
G2048_HD bool quad_is_hot(uint32_t k) {          // all four nibbles of k <= 10
    return ((k & ((k & 0x7777u) + 0x5555u)) & 0x8888u) == 0u;
}
G2048_HD uint32_t quad_base11(uint32_t k) { return (((k >> 12) & 15u) * 11u + ((k >> 8) & 15u)) * 121u + ((k >> 4) & 15u) * 11u + (k & 15u); }
G2048_HD uint32_t quad_place(uint32_t k) { return quad_is_hot(k) ? quad_base11(k) : QUAD_HOT + k; }
// the index whose place is K; false for a hole.  `plain`: no index is hot (every index lives at QUAD_HOT + k; g2048.hip, QuadOrder)
G2048_HD bool quad_unplace(uint32_t K, uint32_t& k, bool plain = false) {
    if (K < QUAD_HOT) {
        if (plain || K >= QUAD_HOT_USED) return false;
        const uint32_t c3 = K % 11u, r2 = K / 11u, c2 = r2 % 11u, r1 = r2 / 11u, c1 = r1 % 11u, c0 = r1 / 11u;
        k = c0 << 12 | c1 << 8 | c2 << 4 | c3;
        return true;
    }
    k = K - QUAD_HOT;
    return k < 65536u && (plain || !quad_is_hot(k));
}

// The records hold the plain 16-bit indices, two per 32-bit word; both halves of a word at once: which nibbles are >= 11, and
// the base-11 value of each half (= its place if it has no such nibble).
G2048_HD uint32_t quad_big_nibbles(uint32_t x) { return x & ((x & 0x77777777u) + 0x55555555u) & 0x88888888u; }
G2048_HD uint32_t quad_base11_halves(uint32_t x) {
    const uint32_t h = (x >> 4) & 0x0F0F0F0Fu;
    uint32_t h4 = h << 2;
#if defined(__HIP_DEVICE_COMPILE__)
    // (the device compiler folds x - h - 4 h (and a b - 135 g) into 64-bit multiply-adds, v_mad_u64_u32: quarter rate.  The empty asm
    // hides that h4 is 4 h; g < 2^24, so 121 g + lo is one full-rate 24-bit multiply-add)
    asm volatile("" : "+v"(h4));
#endif
    const uint32_t b = x - h - h4;                           // every byte 16 hi + lo -> 11 hi + lo
    const uint32_t g = (b >> 8) & 0x00FF00FFu, lo = b & 0x00FF00FFu;
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul24(g, 121u) + lo;                           // every half 256 B1 + B0 -> 121 B1 + B0 (one v_mad_u32_u24)
#else
    return g * 121u + lo;
#endif
}

G2048_HD uint32_t pack16(uint32_t w) {      // bytes b0..b3 (cells 0..3 of a line) -> b0<<12|b1<<8|b2<<4|b3
    return ((w & 0xFu) << 12) | ((w >> 8 & 0xFu) << 8) | ((w >> 16 & 0xFu) << 4) | (w >> 24 & 0xFu);
}

G2048_HD uint32_t nibrev16(uint32_t x) {
    return ((x & 0xFu) << 12) | ((x & 0xF0u) << 4) | ((x >> 4) & 0xF0u) | ((x >> 12) & 0xFu);
}

// rows: row words of the board, cols: column words (transpose)
G2048_HD Packed pack_board(const uint32_t rows[4], const uint32_t cols[4]) {
    Packed p;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        p.R[i] = pack16(rows[i]);
        p.C[i] = pack16(cols[i]);
    }
    return p;
}

G2048_HD Packed pack_board(const Board& b) {
    uint32_t cols[4];
    transpose(b.r, cols);
    return pack_board(b.r, cols);
}

// image g in 0..7 of the dihedral group: bit0 = transpose, bit1 = mirror left-right, bit2 = mirror up-down
// (applied mirror first, transpose last).  The 8 images are the ones QAgent.update visits; its sum over them
// does not depend on their order.
G2048_HD Packed d4_image(const Packed& p, uint32_t g) {
    const bool t = g & 1u, fh = g & 2u, fv = g & 4u;
    Packed q;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t r_src = fv ? p.R[3 - i] : p.R[i];          // up-down mirror reverses the order of rows ...
        uint32_t c_src = fh ? p.C[3 - i] : p.C[i];          // left-right mirror reverses the order of columns
        q.R[i] = fh ? nibrev16(r_src) : r_src;              // ... and a left-right mirror reverses cells inside a row
        q.C[i] = fv ? nibrev16(c_src) : c_src;
    }
    if (t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t tmp = q.R[i];
            q.R[i] = q.C[i];
            q.C[i] = tmp;
        }
    }
    return q;
}

// bit-fields of the packed rows / columns
#define G2048_CELL(p, r, c) (((p).R[r] >> (12 - 4 * (c))) & 0xFu)
#define G2048_PAIR_R(p, r, c) (((p).R[r] >> (8 - 4 * (c))) & 0xFFu)    /* (r,c),(r,c+1) */
#define G2048_PAIR_C(p, c, r) (((p).C[c] >> (8 - 4 * (r))) & 0xFFu)    /* (r,c),(r+1,c) */
#define G2048_TRIP_R(p, r, c) (((p).R[r] >> (4 - 4 * (c))) & 0xFFFu)   /* (r,c),(r,c+1),(r,c+2) */
#define G2048_TRIP_C(p, c, r) (((p).C[c] >> (4 - 4 * (r))) & 0xFFFu)   /* (r,c),(r+1,c),(r+2,c) */

// Flat table slots (offset included) of every feature of one board, in the reference's feature order.
template <int N>
G2048_HD void feature_slots(const Packed& p, uint32_t out[Shape<N>::F]);

// f_2, r_learning.py:17-20: 12 vertical pairs (r-major), 12 horizontal pairs (r-major); 256 slots each
template <>
G2048_HD void feature_slots<2>(const Packed& p, uint32_t out[24]) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) out[4 * r + c] = (uint32_t)(4 * r + c) * 256u + G2048_PAIR_C(p, c, r);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) out[12 + 3 * r + c] = (uint32_t)(12 + 3 * r + c) * 256u + G2048_PAIR_R(p, r, c);
}

// f_3, r_learning.py:24-31: 8 vertical triples, 8 horizontal triples, then for each 2x2 window (3x3 of them)
// the four L-shapes; 4096 slots each
template <>
G2048_HD void feature_slots<3>(const Packed& p, uint32_t out[52]) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) out[4 * r + c] = (uint32_t)(4 * r + c) * 4096u + G2048_TRIP_C(p, c, r);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) out[8 + 2 * r + c] = (uint32_t)(8 + 2 * r + c) * 4096u + G2048_TRIP_R(p, r, c);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int w = 3 * r + c;
            uint32_t tl = G2048_CELL(p, r, c), tr = G2048_CELL(p, r, c + 1), br = G2048_CELL(p, r + 1, c + 1);
            uint32_t bottom = G2048_PAIR_R(p, r + 1, c);            // bl, br
            uint32_t top = G2048_PAIR_R(p, r, c);                   // tl, tr
            uint32_t left = G2048_PAIR_C(p, c, r);                  // tl, bl
            out[16 + w] = (uint32_t)(16 + w) * 4096u + ((bottom << 4) | tr);       // bl, br, tr  (x_ex_00)
            out[25 + w] = (uint32_t)(25 + w) * 4096u + ((tl << 8) | bottom);       // tl, bl, br  (x_ex_01)
            out[34 + w] = (uint32_t)(34 + w) * 4096u + ((top << 4) | br);          // tl, tr, br  (x_ex_10)
            out[43 + w] = (uint32_t)(43 + w) * 4096u + ((left << 4) | tr);         // tl, bl, tr  (x_ex_11)
        }
}

// the 17 four-cell features shared by f_4/f_5/f_6 (r_learning.py:40-44): 4 columns, 4 rows, 9 squares
G2048_HD void quad_slots(const Packed& p, uint32_t out[17]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) out[c] = (uint32_t)c * 65536u + p.C[c];
#pragma unroll
    for (int r = 0; r < 4; ++r) out[4 + r] = (uint32_t)(4 + r) * 65536u + p.R[r];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)     // (r,c),(r+1,c),(r,c+1),(r+1,c+1)
            out[8 + 3 * r + c] = (uint32_t)(8 + 3 * r + c) * 65536u + ((G2048_PAIR_C(p, c, r) << 8) | G2048_PAIR_C(p, c + 1, r));
}

// 4 crosses around the middle cells (r_learning.py:51-52): centre, up, left, down, right
G2048_HD void cross_slots(const Packed& p, uint32_t out[4]) {
#pragma unroll
    for (int r = 1; r < 3; ++r)
#pragma unroll
        for (int c = 1; c < 3; ++c) {
            uint32_t idx = (G2048_CELL(p, r, c) << 16) | (G2048_CELL(p, r - 1, c) << 12) | (G2048_CELL(p, r, c - 1) << 8) |
                           (G2048_CELL(p, r + 1, c) << 4) | G2048_CELL(p, r, c + 1);
            out[2 * (r - 1) + (c - 1)] = CROSS_BASE + (uint32_t)(2 * (r - 1) + (c - 1)) * 1048576u + idx;
        }
}

// 12 six-cell features in base 14 on min(tile, 13) (r_learning.py:63-68): 6 tall 3x2 blocks, 6 wide 2x3 blocks
G2048_HD void hex_slots(const Packed& p, uint32_t out[12]) {
    uint32_t y[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            uint32_t v = G2048_CELL(p, r, c);
            y[r][c] = v > 13u ? 13u : v;
        }
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            uint32_t a = 196u * y[r][c] + 14u * y[r + 1][c] + y[r + 2][c];
            uint32_t b = 196u * y[r][c + 1] + 14u * y[r + 1][c + 1] + y[r + 2][c + 1];
            out[3 * r + c] = HEX_BASE + (uint32_t)(3 * r + c) * HEX_SIZE + 2744u * a + b;
        }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            uint32_t a = 196u * y[r][c] + 14u * y[r][c + 1] + y[r][c + 2];
            uint32_t b = 196u * y[r + 1][c] + 14u * y[r + 1][c + 1] + y[r + 1][c + 2];
            out[6 + 2 * r + c] = HEX_BASE + (uint32_t)(6 + 2 * r + c) * HEX_SIZE + 2744u * a + b;
        }
}

// ---- where an entry of a six-cell (base-14) table lives in memory: the same idea as table_place.  Index k = sum d_p 14^p
// goes to 64 * (sum (d_p >> 1) 7^p) + sum (d_p & 1) 2^p — a bijection of [0, 14^6) (7^6 * 64 = 14^6): the six low digit
// bits select the entry inside a 256-byte block and the block is chosen by the digits' upper parts, of which small tiles
// have few (tiles <= 32: 3^6 = 729 blocks per table instead of 6^5 = 7 776 lines).
G2048_HD uint32_t hex_place(uint32_t k) {
    uint32_t hi = 0, lo = 0, m7 = 1;
    for (uint32_t p = 0; p < 6u; ++p) {
        const uint32_t d = k % 14u;
        k /= 14u;
        hi += (d >> 1) * m7;
        lo |= (d & 1u) << p;
        m7 *= 7u;
    }
    return 64u * hi + lo;
}

// the inverse of hex_place
G2048_HD uint32_t hex_unplace(uint32_t j) {
    uint32_t hi = j >> 6, k = 0, m14 = 1;
    for (uint32_t p = 0; p < 6u; ++p) {
        k += (2u * (hi % 7u) + ((j >> p) & 1u)) * m14;
        hi /= 7u;
        m14 *= 14u;
    }
    return k;
}

// a slot of an n >= 4 table -> its place in memory
G2048_HD uint32_t table_place_any(uint32_t slot) {
    if (slot < HEX_BASE) return table_place(slot);
    const uint32_t t = (slot - HEX_BASE) / HEX_SIZE, k = slot - HEX_BASE - t * HEX_SIZE;
    return HEX_BASE + t * HEX_SIZE + hex_place(k);
}

// hex_slots with every entry already at its place: table_place_any(hex_slots()[j]) without the digit extraction
G2048_HD void hex_slots_placed(const Packed& p, uint32_t out[12]) {
    uint32_t h[4][4], l[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            uint32_t v = G2048_CELL(p, r, c);
            v = v > 13u ? 13u : v;
            h[r][c] = v >> 1;
            l[r][c] = v & 1u;
        }
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const uint32_t a = 49u * h[r][c] + 7u * h[r + 1][c] + h[r + 2][c], la = l[r][c] << 2 | l[r + 1][c] << 1 | l[r + 2][c];
            const uint32_t b = 49u * h[r][c + 1] + 7u * h[r + 1][c + 1] + h[r + 2][c + 1], lb = l[r][c + 1] << 2 | l[r + 1][c + 1] << 1 | l[r + 2][c + 1];
            out[3 * r + c] = HEX_BASE + (uint32_t)(3 * r + c) * HEX_SIZE + 64u * (343u * a + b) + (la << 3 | lb);
        }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const uint32_t a = 49u * h[r][c] + 7u * h[r][c + 1] + h[r][c + 2], la = l[r][c] << 2 | l[r][c + 1] << 1 | l[r][c + 2];
            const uint32_t b = 49u * h[r + 1][c] + 7u * h[r + 1][c + 1] + h[r + 1][c + 2], lb = l[r + 1][c] << 2 | l[r + 1][c + 1] << 1 | l[r + 1][c + 2];
            out[6 + 2 * r + c] = HEX_BASE + (uint32_t)(6 + 2 * r + c) * HEX_SIZE + 64u * (343u * a + b) + (la << 3 | lb);
        }
}

template <>
G2048_HD void feature_slots<4>(const Packed& p, uint32_t out[17]) { quad_slots(p, out); }

template <>
G2048_HD void feature_slots<5>(const Packed& p, uint32_t out[21]) {
    quad_slots(p, out);
    cross_slots(p, out + 17);
}

template <>
G2048_HD void feature_slots<6>(const Packed& p, uint32_t out[33]) {
    quad_slots(p, out);
    cross_slots(p, out + 17);
    hex_slots(p, out + 21);
}

// ---- feature_slots at their places in memory: what the kernels that read or write the table use (n = 2, 3: index order)
// The four- and five-cell features are computed IN the transposed domain instead of transposing 21 indices one by one:
// only the eight line indices are transposed (two per 32-bit word: no shift of the network crosses a half), and in a
// transposed line — [bit 3 of its four cells | bit 2 | bit 1 | bit 0], cell j at bit 3 - j of each group — the bits of two
// neighbouring cells are neighbours, so a square or a cross is two masked shifts of its lines.  (~95 VALU instructions per
// board instead of ~240; tests/test_hostcheck_logic.py holds it to table_place(feature_slots).)
G2048_HD uint32_t bit_transpose16x2(uint32_t x) {     // both 16-bit halves transposed
    uint32_t t = (x ^ (x >> 3)) & 0x0A0A0A0Au;
    x ^= t ^ (t << 3);
    t = (x ^ (x >> 6)) & 0x00CC00CCu;
    x ^= t ^ (t << 6);
    return x;
}
// the packed lines of a board two to a word: RR[k] = R[2k] | R[2k + 1] << 16, CC likewise
struct PackedPairs {
    uint32_t RR[2], CC[2];
};
// pack16 of two byte-words at once: (w << 4 | w >> 8) has b0 << 4 | b1 in byte 0 and b2 << 4 | b3 in byte 2, and one byte
// permute puts the four bytes of two lines in place (5 instructions for two lines instead of 14)
G2048_HD uint32_t pack16x2(uint32_t w0, uint32_t w1) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t u0 = (w0 << 4) | (w0 >> 8), u1 = (w1 << 4) | (w1 >> 8);
    return __builtin_amdgcn_perm(u1, u0, 0x04060002u);      // bytes, low to high: u0.2, u0.0, u1.2, u1.0
#else
    return pack16(w0) | pack16(w1) << 16;
#endif
}
G2048_HD PackedPairs pack_pairs(const uint32_t rows[4], const uint32_t cols[4]) {
    PackedPairs q;
    q.RR[0] = pack16x2(rows[0], rows[1]);
    q.RR[1] = pack16x2(rows[2], rows[3]);
    q.CC[0] = pack16x2(cols[0], cols[1]);
    q.CC[1] = pack16x2(cols[2], cols[3]);
    return q;
}
G2048_HD PackedPairs pack_pairs(const Board& b) {
    uint32_t cols[4];
    transpose(b.r, cols);
    return pack_pairs(b.r, cols);
}
G2048_HD PackedPairs pair_up(const Packed& p) {
    return PackedPairs{{p.R[0] | p.R[1] << 16, p.R[2] | p.R[3] << 16}, {p.C[0] | p.C[1] << 16, p.C[2] | p.C[3] << 16}};
}
G2048_HD Packed unpair(const PackedPairs& q) {
    Packed p;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        p.R[i] = (i & 1) ? q.RR[i >> 1] >> 16 : q.RR[i >> 1] & 0xFFFFu;
        p.C[i] = (i & 1) ? q.CC[i >> 1] >> 16 : q.CC[i >> 1] & 0xFFFFu;
    }
    return p;
}

struct PlacedLines {
    uint32_t R[4], C[4];        // table_place of the packed rows / columns
};
G2048_HD PlacedLines place_lines(const PackedPairs& q) {
    PlacedLines t;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const uint32_t r = bit_transpose16x2(q.RR[k]), c = bit_transpose16x2(q.CC[k]);
        t.R[2 * k] = r & 0xFFFFu;
        t.R[2 * k + 1] = r >> 16;
        t.C[2 * k] = c & 0xFFFFu;
        t.C[2 * k + 1] = c >> 16;
    }
    return t;
}
// table_place(quad_slots(p)[f]) - f * 65536: the place inside the feature's own table
G2048_HD void quad_places(const PlacedLines& t, uint32_t out[17]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) out[c] = t.C[c];
#pragma unroll
    for (int r = 0; r < 4; ++r) out[4 + r] = t.R[r];
#pragma unroll
    for (int r = 0; r < 3; ++r) {       // square (r, c) = cells r, r + 1 of columns c and c + 1: bits 3 - r, 2 - r of every group
        uint32_t m[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) m[c] = (t.C[c] >> (2 - r)) & 0x3333u;
#pragma unroll
        for (int c = 0; c < 3; ++c) out[8 + 3 * r + c] = m[c] << 2 | m[c + 1];
    }
}
// table_place(cross_slots(p)[j]) - CROSS_BASE - j * 1048576 (centre << 16 | up, left, down, right transposed)
G2048_HD void cross_places(const PackedPairs& q, const PlacedLines& t, uint32_t out[4]) {
#pragma unroll
    for (int r = 1; r < 3; ++r)
#pragma unroll
        for (int c = 1; c < 3; ++c) {
            const uint32_t ud = (t.C[c] >> (2 - r)) & 0x5555u;      // up at bit 2, down at bit 0 of every group
            const uint32_t lr = (t.R[r] >> (2 - c)) & 0x5555u;      // left at bit 2, right at bit 0
            const uint32_t centre = (q.RR[r >> 1] >> (16 * (r & 1) + 12 - 4 * c)) & 0xFu;
            out[2 * (r - 1) + (c - 1)] = centre << 16 | ud << 1 | lr;
        }
}

template <int N>
G2048_HD void memory_slots(const PackedPairs& q, uint32_t* out) {
    if constexpr (N < 4) {
        feature_slots<N>(unpair(q), out);
    } else {
        const PlacedLines t = place_lines(q);
        quad_places(t, out);
#pragma unroll
        for (int f = 0; f < 17; ++f) out[f] += (uint32_t)f * 65536u;
        if constexpr (N >= 5) {
            cross_places(q, t, out + 17);
#pragma unroll
            for (int j = 0; j < 4; ++j) out[17 + j] += CROSS_BASE + (uint32_t)j * 1048576u;
        }
        if constexpr (N == 6) hex_slots_placed(unpair(q), out + 21);
    }
}
template <int N>
G2048_HD void memory_slots(const Packed& p, uint32_t* out) {
    if constexpr (N < 4)
        feature_slots<N>(p, out);
    else
        memory_slots<N>(pair_up(p), out);
}

// first slot of feature i (host-side layout queries)
G2048_HD constexpr uint32_t feature_offset(int n, int i) {
    switch (n) {
        case 2: return (uint32_t)i * 256u;
        case 3: return (uint32_t)i * 4096u;
        case 4: return (uint32_t)i * 65536u;
        default:
            if (i < 17) return (uint32_t)i * 65536u;
            if (i < 21) return CROSS_BASE + (uint32_t)(i - 17) * 1048576u;
            return HEX_BASE + (uint32_t)(i - 21) * HEX_SIZE;
    }
}

G2048_HD constexpr uint32_t feature_size(int n, int i) {
    switch (n) {
        case 2: return 256u;
        case 3: return 4096u;
        case 4: return 65536u;
        default: return i < 17 ? 65536u : (i < 21 ? 1048576u : HEX_SIZE);
    }
}

// ---- symmetry orbits of the update (g2048.hip: LDS-owner kernels, k_td_play's index records, k_td_update_tail)
// n >= 4: one variant per LDS-owned orbit, encoding the orbit's representative feature (outer line 0, inner line 1,
// corner square 8, edge square 9, centre square 12, cross 17; find_orbits checks that these are the representatives)
constexpr int ORBIT_REPS[6] = {0, 1, 8, 9, 12, 17};
// images (bit g = d4_image g) that the owner kernel visits for each representative: one per coset of its stabiliser
// (columns 0 / 1: up-down mirror; corner square and cross: transpose; edge square: left-right mirror; the centre square
// is fixed by the whole group).  4 + 4 + 4 + 4 + 1 + 4 = 21 LDS adds per record instead of 48.  find_orbits verifies
// these masks against the brute-force enumeration of all 8 images.
constexpr uint32_t COSET_MASK[6] = {0x27u, 0x27u, 0x55u, 0x1Bu, 0x01u, 0x55u};
// the two f_6 orbits (k_td_update_tail): the corner blocks' representative (feature 21) is fixed by nothing, the middle
// blocks' (feature 22) by the left-right mirror
constexpr uint32_t HEX_COSET_MASK[2] = {0xFFu, 0x1Bu};
// n = 2, 3 (round 3): the same reduction.  f_2's 24 pairs fall into 4 orbits (the 8 border pairs that touch a corner, the
// 8 pairs that run from the border inwards, the 4 middle border pairs, the 4 pairs of the centre square); f_3's 52 triples
// into 8 (border lines, inner lines, and the L-shapes of the corner / edge / centre windows by which cell they leave out).
// A record then costs 24 (n = 2) or 52 (n = 3) LDS adds instead of 8 x 24 = 192 or 8 x 52 = 416, into orbit tables of
// 4 x 256 or 8 x 4 096 slots.  Representatives and coset masks were found with the host build of these headers
// (tests/test_hostcheck_logic.py repeats the search) and find_orbits checks them at context creation.
template <int N> struct SmallOrbits;
template <> struct SmallOrbits<2> {
    static constexpr int COUNT = 4, PER_CHUNK = 4;
    static constexpr uint32_t SIZE = 256u;
    static constexpr int rep(int o) { constexpr int R[4] = {0, 1, 4, 5}; return R[o]; }
    static constexpr uint32_t mask(int o) { constexpr uint32_t M[4] = {0xFFu, 0xFFu, 0x27u, 0x27u}; return M[o]; }
};
template <> struct SmallOrbits<3> {
    static constexpr int COUNT = 8, PER_CHUNK = 4;
    static constexpr uint32_t SIZE = 4096u;
    static constexpr int rep(int o) { constexpr int R[8] = {0, 1, 16, 17, 18, 20, 21, 24}; return R[o]; }
    static constexpr uint32_t mask(int o) { constexpr uint32_t M[8] = {0xFFu, 0xFFu, 0x55u, 0xFFu, 0xFFu, 0x55u, 0xFFu, 0x55u}; return M[o]; }
};

// the j-th visited image of orbit variant V
constexpr uint32_t coset_rank(uint32_t mask, uint32_t g) {
    uint32_t r = 0;
    for (uint32_t b = 0; b < g; ++b) r += (mask >> b) & 1u;
    return r;
}

}  // namespace g2048
