// dca_common.h — shared device/host helpers for libdca_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dca.h"
#include "../../include/dca_debug.h"

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

// ------------------------------------------------------------------------------------------
// cube4 (the 4x4x4 cube of the reference's C++ core, cpp/environments.cpp:263-370; 96 stickers = face*16 + row*4 + col, faces
// U D L R B F; 24 moves: a = 2 f + d turns the OUTER layer of face f, a = 12 + 2 f + d the INNER slice next to it; d = 0 is
// the reference's "_n1" direction, d = 1 its inverse).  next[i] = cur[p[a][i]], like cube3.  Built at compile time from the
// geometry: a "_n1" turn rotates the face's own 4x4 grid — new(r, c) = old(c, 3 - r), outer moves only — and carries four
// strips of four stickers around the face in 4-cycles (c0 c1 c2 c3): next[c0] = cur[c1], next[c1] = cur[c2], ...; strip k of
// a layer is {start + k * step}, the inner slice's strips are the outer ones shifted by `inner`.  Pinned against the
// reference's own compiled class (oracle/_ref) on every move: sha256 7eb3d1a8... of the 24 x 96 table (tests).
// ------------------------------------------------------------------------------------------
struct Cube4Perm {
    uint8_t p[24][96];
};

constexpr Cube4Perm make_cube4_perm() {
    // per face: first strip's four stickers (one per neighbouring face), the step from strip to strip, the shift to the inner slice
    constexpr int side[6][3][4] = {
        {{35, 67, 51, 83}, {4, 4, 4, 4}, {-1, -1, -1, -1}},   // U
        {{32, 80, 48, 64}, {4, 4, 4, 4}, {1, 1, 1, 1}},       // D
        {{0, 80, 16, 79}, {1, 1, 1, -1}, {4, 4, 4, -4}},      // L
        {{12, 67, 28, 92}, {1, -1, 1, 1}, {-4, 4, -4, -4}},   // R
        {{3, 32, 28, 63}, {4, 1, -4, -1}, {-1, 4, 1, -4}},    // B
        {{0, 51, 31, 44}, {4, -1, -4, 1}, {1, 4, -1, -4}},    // F
    };
    Cube4Perm t{};
    for (int a = 0; a < 24; a++)
        for (int i = 0; i < 96; i++) t.p[a][i] = (uint8_t)i;
    for (int f = 0; f < 6; f++)
        for (int layer = 0; layer < 2; layer++) {
            const int a = layer * 12 + 2 * f;
            if (layer == 0)
                for (int r = 0; r < 4; r++)
                    for (int c = 0; c < 4; c++) {
                        const int dst = f * 16 + r * 4 + c, src = f * 16 + c * 4 + (3 - r);
                        t.p[a][dst] = (uint8_t)src;
                        t.p[a + 1][src] = (uint8_t)dst;
                    }
            for (int k = 0; k < 4; k++) {
                int cyc[4] = {0, 0, 0, 0};
                for (int j = 0; j < 4; j++) cyc[j] = side[f][0][j] + k * side[f][1][j] + layer * side[f][2][j];
                for (int j = 0; j < 4; j++) {
                    const int x = cyc[j], y = cyc[(j + 1) % 4];
                    t.p[a][x] = (uint8_t)y;
                    t.p[a + 1][y] = (uint8_t)x;
                }
            }
        }
    return t;
}
inline constexpr Cube4Perm kCube4Perm = make_cube4_perm();

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
    return (env == DCA_ENV_CUBE3 || env == DCA_ENV_CUBE4) ? (uint32_t)i : env == DCA_ENV_NPUZZLE ? (uint32_t)((i + 1) % D) : 0u;
}
// is_solved, one byte at a time: `ok` after byte i (value b) of a state given `ok` before it.  cube3 / puzzles / lightsout:
// equality with the goal byte (cube3.py:71-75, cpp:119-126,249-256).  cube4: every face shows ONE colour (sticker / 16 equal
// to the face's first sticker / 16 — cpp/environments.cpp:355-365: any arrangement of same-coloured stickers counts);
// `first` carries the current face's first colour between calls.
__host__ __device__ inline bool solved_step(int env, int D, int i, uint32_t b, bool ok, uint32_t& first) {
    if (env == DCA_ENV_CUBE4) {
        if ((i & 15) == 0) first = b >> 4;
        return ok && (b >> 4) == first;
    }
    return ok && b == goal_byte(env, D, i);
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
