// TEST-ONLY host build of the device headers (2048_amd/csrc/board_ops.hpp, features.hpp).
// Lets `pytest -m "not gpu"` check the kernels' integer logic against the oracle in a container that has
// no GPU.  It is not a product path: nothing under 2048_amd/ or game2048/ loads this library.
#include <stdint.h>
#include <string.h>

#include "../../2048_amd/csrc/features.hpp"

using namespace g2048;

static Board load(const uint8_t* p) {
    Board b;
    memcpy(b.r, p, 16);
    return b;
}
static void store(uint8_t* p, const Board& b) { memcpy(p, b.r, 16); }

template <int D>
static void one_dir(const Board& b, const uint32_t cols[4], uint8_t* after, int32_t* reward, uint8_t* changed) {
    Moved m = move_dir<D>(b.r, cols);
    store(after + 16 * D, m.after);
    reward[D] = (int32_t)merged_score(m.ma, m.mb);
    if (m.changed) *changed |= (uint8_t)(1u << D);
}

extern "C" {

void hc_move_all(const uint8_t* boards, int64_t count, uint8_t* after, int32_t* reward, uint8_t* changed) {
    for (int64_t i = 0; i < count; ++i) {
        Board b = load(boards + 16 * i);
        uint32_t cols[4];
        transpose(b.r, cols);
        changed[i] = 0;
        one_dir<0>(b, cols, after + 64 * i, reward + 4 * i, changed + i);
        one_dir<1>(b, cols, after + 64 * i, reward + 4 * i, changed + i);
        one_dir<2>(b, cols, after + 64 * i, reward + 4 * i, changed + i);
        one_dir<3>(b, cols, after + 64 * i, reward + 4 * i, changed + i);
    }
}

void hc_terminal(const uint8_t* boards, int64_t count, uint8_t* over, uint8_t* n_empty, uint8_t* n_pairs, uint8_t* top) {
    for (int64_t i = 0; i < count; ++i) {
        Board b = load(boards + 16 * i);
        over[i] = game_over(b);
        n_empty[i] = (uint8_t)empty_count(b);
        n_pairs[i] = (uint8_t)adjacent_pairs(b);
        top[i] = (uint8_t)max_tile(b);
    }
}

void hc_spawn_injected(uint8_t* boards, int64_t count, const uint8_t* r10, const uint8_t* k) {
    for (int64_t i = 0; i < count; ++i) {
        Board b = load(boards + 16 * i);
        place_tile(b, r10[i], k[i], empty_bits(b));
        store(boards + 16 * i, b);
    }
}

// draws[count][ndraws] raw 64-bit outputs of each lane's stream
void hc_rng_stream(uint64_t seed, uint64_t lane0, int64_t count, int ndraws, uint64_t* draws, uint64_t* state_out) {
    for (int64_t i = 0; i < count; ++i) {
        Rng g = seed_lane(seed, lane0 + (uint64_t)i);
        for (int j = 0; j < ndraws; ++j) draws[i * ndraws + j] = next_u64(g);
        state_out[2 * i] = g.s0;
        state_out[2 * i + 1] = g.s1;
    }
}

void hc_new_games(uint64_t seed, uint64_t lane0, int64_t count, uint8_t* boards) {
    for (int64_t i = 0; i < count; ++i) {
        Rng g = seed_lane(seed, lane0 + (uint64_t)i);
        store(boards + 16 * i, new_game(g));
    }
}

// the orbit representatives and coset masks the update kernels use: reps[8] (6 LDS-owned + the two f_6), masks[8]
void hc_coset_masks(int32_t* reps, uint32_t* masks) {
    for (int v = 0; v < 6; ++v) {
        reps[v] = ORBIT_REPS[v];
        masks[v] = COSET_MASK[v];
    }
    reps[6] = 21; reps[7] = 22;
    masks[6] = HEX_COSET_MASK[0]; masks[7] = HEX_COSET_MASK[1];
}

// n = 2, 3: the orbit representatives and coset masks the LDS-owner update uses (SmallOrbits<N>, features.hpp)
int hc_small_orbits(int n, int32_t* reps, uint32_t* masks) {
    if (n == 2) {
        for (int o = 0; o < SmallOrbits<2>::COUNT; ++o) { reps[o] = SmallOrbits<2>::rep(o); masks[o] = SmallOrbits<2>::mask(o); }
        return SmallOrbits<2>::COUNT;
    }
    if (n == 3) {
        for (int o = 0; o < SmallOrbits<3>::COUNT; ++o) { reps[o] = SmallOrbits<3>::rep(o); masks[o] = SmallOrbits<3>::mask(o); }
        return SmallOrbits<3>::COUNT;
    }
    return -1;
}

// out[count][8][F] flat slots of every feature of every D4 image
int hc_image_slots(int n, const uint8_t* boards, int64_t count, int32_t* out) {
    for (int64_t i = 0; i < count; ++i) {
        Packed p = pack_board(load(boards + 16 * i));
        for (uint32_t g = 0; g < 8; ++g) {
            Packed q = d4_image(p, g);
            uint32_t s[52];
            int F;
            switch (n) {
                case 2: feature_slots<2>(q, s); F = 24; break;
                case 3: feature_slots<3>(q, s); F = 52; break;
                case 4: feature_slots<4>(q, s); F = 17; break;
                case 5: feature_slots<5>(q, s); F = 21; break;
                case 6: feature_slots<6>(q, s); F = 33; break;
                default: return -1;
            }
            for (int f = 0; f < F; ++f) out[(i * 8 + g) * F + f] = (int32_t)s[f];
        }
    }
    return 0;
}
// where table slots live in memory (table_place_any) and memory_slots<n> of boards: out[count][F]
void hc_table_place(const uint32_t* slots, int64_t count, uint32_t* out) {
    for (int64_t i = 0; i < count; ++i) out[i] = table_place_any(slots[i]);
}
void hc_hex_place(uint32_t first, int64_t count, uint32_t* out) {
    for (int64_t i = 0; i < count; ++i) out[i] = hex_place(first + (uint32_t)i);
}
int hc_memory_slots(int n, const uint8_t* boards, int64_t count, int32_t* out) {
    for (int64_t i = 0; i < count; ++i) {
        Packed p = pack_board(load(boards + 16 * i));
        uint32_t s[52];
        int F;
        switch (n) {
            case 2: memory_slots<2>(p, s); F = 24; break;
            case 3: memory_slots<3>(p, s); F = 52; break;
            case 4: memory_slots<4>(p, s); F = 17; break;
            case 5: memory_slots<5>(p, s); F = 21; break;
            case 6: memory_slots<6>(p, s); F = 33; break;
            default: return -1;
        }
        for (int f = 0; f < F; ++f) out[i * F + f] = (int32_t)s[f];
    }
    return 0;
}
// the four-cell orbits' table order: place[k], hot[k] for k < 65536; unplace[K] (-1: hole) for K < QUAD_DSIZE; and the two-at-a-time
// forms on the word k | k2 << 16 for `pairs` pairs (k, k2)
int hc_quad_place(uint32_t* place, uint8_t* hot, int32_t* unplace, int64_t pairs, const uint32_t* words, uint32_t* big, uint32_t* b11) {
    for (uint32_t k = 0; k < 65536u; ++k) {
        place[k] = quad_place(k);
        hot[k] = quad_is_hot(k) ? 1 : 0;
    }
    for (uint32_t K = 0; K < QUAD_DSIZE; ++K) {
        uint32_t k = 0;
        unplace[K] = quad_unplace(K, k) ? (int32_t)k : -1;
    }
    for (int64_t i = 0; i < pairs; ++i) {
        big[i] = quad_big_nibbles(words[i]);
        b11[i] = quad_base11_halves(words[i]);
    }
    return (int)QUAD_DSIZE;
}
void hc_cross_order(int64_t count, uint32_t* fwd, uint32_t* back) {       // cross_order(k) and cross_unorder(k) for k = 0 .. count-1
    for (int64_t k = 0; k < count; ++k) {
        fwd[k] = cross_order((uint32_t)k);
        back[k] = cross_unorder((uint32_t)k);
    }
}
}
