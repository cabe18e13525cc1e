// dca_common.h — shared device/host helpers for libdca_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dca.h"

namespace dca {

// ------------------------------------------------------------------------------------------
// cube3 move tables, built at compile time from the 4-cycles of the six "-1" face turns
// (sticker = face*9 + row*3 + col, faces U D L R B F).  (a b c d) under a "-1" turn:
// next[a]=cur[b], next[b]=cur[c], next[c]=cur[d], next[d]=cur[a]; "+1" is the inverse.
// Equals Cube3._compute_rotation_idxs (environments/cube3.py:183-256) and the literal
// rotateIdxs tables of cpp/environments.h:75-105 (pinned: sha256 d090eb61… in the tests).
// ------------------------------------------------------------------------------------------
struct Cube3Perm {
    uint8_t p[12][54];
};

constexpr Cube3Perm make_cube3_perm() {
    constexpr int cyc[6][5][4] = {
        {{0, 2, 8, 6}, {1, 5, 7, 3}, {20, 38, 29, 47}, {23, 41, 32, 50}, {26, 44, 35, 53}},
        {{9, 11, 17, 15}, {10, 14, 16, 12}, {18, 45, 27, 36}, {21, 48, 30, 39}, {24, 51, 33, 42}},
        {{0, 45, 9, 44}, {1, 46, 10, 43}, {2, 47, 11, 42}, {18, 20, 26, 24}, {19, 23, 25, 21}},
        {{6, 38, 15, 51}, {7, 37, 16, 52}, {8, 36, 17, 53}, {27, 29, 35, 33}, {28, 32, 34, 30}},
        {{2, 18, 15, 35}, {5, 19, 12, 34}, {8, 20, 9, 33}, {36, 38, 44, 42}, {37, 41, 43, 39}},
        {{0, 29, 17, 24}, {3, 28, 14, 25}, {6, 27, 11, 26}, {45, 47, 53, 51}, {46, 50, 52, 48}},
    };
    Cube3Perm t{};
    for (int a = 0; a < 12; a++)
        for (int i = 0; i < 54; i++) t.p[a][i] = (uint8_t)i;
    for (int f = 0; f < 6; f++)
        for (int c = 0; c < 5; c++)
            for (int j = 0; j < 4; j++) {
                int a = cyc[f][c][j], b = cyc[f][c][(j + 1) % 4];
                t.p[2 * f][a] = (uint8_t)b;
                t.p[2 * f + 1][b] = (uint8_t)a;
            }
    return t;
}
inline constexpr Cube3Perm kCube3Perm = make_cube3_perm();

// blank-swap target for the sliding puzzles (n_puzzle.py:174-214 / environments.cpp:4-46):
// moves U,D,L,R; ineligible moves are no-ops (return z).
__host__ __device__ inline int npuzzle_swap(int dim, int z, int a) {
    int i = z / dim, j = z - i * dim;
    switch (a) {
        case 0: return i < dim - 1 ? z + dim : z;
        case 1: return i > 0 ? z - dim : z;
        case 2: return j < dim - 1 ? z + 1 : z;
        default: return j > 0 ? z - 1 : z;
    }
}

// LightsOut: does pressing cell `a` flip cell `i`?  (move matrix lights_out.py:33-44 / environments.cpp:133-154:
// the cell itself, a + dim if its x = a / dim < dim - 1, a - dim if x > 0, a + 1 if its y = a % dim < dim - 1, a - 1 if y > 0)
__host__ __device__ inline uint32_t lightsout_flip(int dim, int a, int i) {
    const int x = a / dim, y = a - x * dim;
    return (uint32_t)((i == a) | (x < dim - 1 && i == a + dim) | (x > 0 && i == a - dim) | (y < dim - 1 && i == a + 1) |
                      (y > 0 && i == a - 1));
}

// byte i of the goal state: cube3 arange(54) (cube3.py:62-69), puzzles 1..n^2-1,0 (n_puzzle.py:69-76), lightsout zeros
// (lights_out.py:55-63)
__host__ __device__ inline uint32_t goal_byte(int env, int D, int i) {
    return env == DCA_ENV_CUBE3 ? (uint32_t)i : env == DCA_ENV_NPUZZLE ? (uint32_t)((i + 1) % D) : 0u;
}

// ------------------------------------------------------------------------------------------
// library state hash (include/dca.h)
// ------------------------------------------------------------------------------------------
constexpr uint64_t kHashSeed = 0x9E3779B97F4A7C15ull;
constexpr uint64_t kHashDimMul = 0xD6E8FEB86659FD93ull;
constexpr uint64_t kHashMul = 0xFF51AFD7ED558CCDull;
constexpr uint64_t kHashFin = 0xC4CEB9FE1A85EC53ull;

__host__ __device__ inline uint64_t hash_init(int D) { return kHashSeed ^ ((uint64_t)D * kHashDimMul); }
__host__ __device__ inline uint64_t hash_word(uint64_t h, uint64_t w) {
    h ^= w;
    h *= kHashMul;
    h ^= h >> 32;
    return h;
}
__host__ __device__ inline uint64_t hash_final(uint64_t h) {
    h ^= h >> 33;
    h *= kHashFin;
    h ^= h >> 33;
    return h;
}

// Manhattan distance contribution of tile `t` sitting at position `pos` of a dim x dim sliding puzzle
// (goal position of tile t is t-1; the blank does not count)
__host__ __device__ inline uint32_t manhattan_term(int dim, uint32_t pos, uint32_t t) {
    if (t == 0) return 0;
    int g = (int)t - 1;
    int dr = (int)(pos / dim) - g / dim, dc = (int)(pos % dim) - g % dim;
    return (uint32_t)((dr < 0 ? -dr : dr) + (dc < 0 ? -dc : dc));
}

// built-in heuristics (include/dca.h DCA_HEUR_*); `sum` = sum_i s_i*(7i+3), `h` = state hash, `manh` = Manhattan distance
__host__ __device__ inline float heur_from(int id, uint64_t sum, uint64_t h, uint32_t manh = 0) {
    switch (id) {
        case DCA_HEUR_MOD97: return (float)(sum % 97) / 50.0f;
        case DCA_HEUR_KNUTH3: {
            uint64_t x = (sum * 2654435761ull) & 0xFFFFFFFFull;
            return (float)((double)x / 4294967296.0 * 3.0);
        }
        case DCA_HEUR_HASHU01: return (float)(10.0 + 5.0 * ((double)(h >> 11) / 9007199254740992.0));
        case DCA_HEUR_MANHATTAN: return (float)manh;
        default: return 0.0f;
    }
}

// ------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------
void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);

#define DCA_HIP(call)                                     \
    do {                                                  \
        hipError_t _e = (call);                           \
        if (_e != hipSuccess) return ::dca::hip_fail(_e, #call); \
    } while (0)

#define DCA_ARG(cond)                                        \
    do {                                                     \
        if (!(cond)) {                                       \
            ::dca::set_error("bad argument: %s", #cond);     \
            return DCA_E_BADARG;                             \
        }                                                    \
    } while (0)

inline int launch_check(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, what);
    return 0;
}

}  // namespace dca
