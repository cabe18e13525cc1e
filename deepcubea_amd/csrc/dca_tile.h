// dca_tile.h — LDS parent-tile helpers shared by the stand-alone expansion kernels (dca_env.hip) and the
// engine's fused pop->expand kernel (dca_engine.hip).
#pragma once
#include "dca_common.h"

namespace dca {

constexpr int kTileParents = 64;
constexpr int kThreads = 256;

template <int ENV, int DIM>
struct EnvT;
template <>
struct EnvT<DCA_ENV_CUBE3, 0> {
    static constexpr int D = 54, A = 12, DEPTH = 6;
};
template <int DIM>
struct EnvT<DCA_ENV_NPUZZLE, DIM> {
    static constexpr int D = DIM * DIM, A = 4, DEPTH = DIM * DIM;
};

template <>
struct EnvT<DCA_ENV_CUBE4, 0> {
    static constexpr int D = 96, A = 24, DEPTH = 6;  // (no network for cube4 in the reference: DEPTH only sizes unused code)
};

template <int DIM>
struct EnvT<DCA_ENV_LIGHTSOUT, DIM> {
    static constexpr int D = DIM * DIM, A = DIM * DIM, DEPTH = 6;  // (ResnetModel(num_tiles, 6, ...), lights_out.py:80)
};

// LDS view of one parent tile + its move tables
template <int ENV, int DIM, int TP = kTileParents>
struct Tile {
    using E = EnvT<ENV, DIM>;
    static constexpr int PAR_BYTES = ((TP * E::D + 15) / 16) * 16;
    static constexpr int TAB_BYTES = ENV == DCA_ENV_CUBE3 ? ((12 * 54 + 15) / 16) * 16
                                     : ENV == DCA_ENV_CUBE4 ? 24 * 96 : ENV == DCA_ENV_NPUZZLE ? TP * 8 : 16;
    static constexpr int LDS_BYTES = PAR_BYTES + TAB_BYTES + 16 + 256;  // slack: the last one-hot lane may peek one row past the tile

    const uint8_t* par;  // [64][D]
    const uint8_t* tab;  // cube3: perm[12*54]; puzzle: per parent {z, s0, s1, s2, s3, pad..} (8 B)

    // byte i of child (parent r, move a)
    __device__ __forceinline__ uint32_t child_byte(uint32_t r, uint32_t a, uint32_t i) const {
        if constexpr (ENV == DCA_ENV_CUBE3 || ENV == DCA_ENV_CUBE4) {
            return par[r * E::D + tab[a * E::D + i]];
        } else if constexpr (ENV == DCA_ENV_LIGHTSOUT) {
            // (state + 1) % 2 on the pressed cell and its neighbours (lights_out.py:161 / environments.cpp:172-174)
            const uint32_t v = par[r * E::D + i];
            return lightsout_flip(DIM, (int)a, (int)i) ? ((v + 1u) & 1u) : v;
        } else {
            uint32_t z = tab[r * 8];
            uint32_t s = tab[r * 8 + 1 + a];
            // next[z] = cur[s]; next[s] = 0   (n_puzzle.py:226-227; s == z is a no-op move)
            uint32_t src = (i == z) ? s : i;
            uint32_t v = par[r * E::D + src];
            return (i == s) ? 0u : v;
        }
    }
    // network-input byte (cube3.py:77-85: sticker // 9; puzzles: the tile itself)
    __device__ __forceinline__ uint32_t nnet_byte(uint32_t r, uint32_t a, uint32_t i) const {
        uint32_t b = child_byte(r, a, i);
        if constexpr (ENV == DCA_ENV_CUBE3) return (b * 57u) >> 9;  // == b / 9 for b < 64
        return b;
    }
};

// device copy of the gather map (constant-initialised from the same constexpr builder)
static __constant__ Cube3Perm d_cube3_perm = make_cube3_perm();
static __constant__ Cube4Perm d_cube4_perm = make_cube4_perm();

__device__ __forceinline__ void stage_tile(uint8_t* lds, const uint8_t* __restrict__ g, uint32_t nbytes, bool aligned) {
    if (aligned) {
        uint32_t nch = nbytes >> 4;
        for (uint32_t q = threadIdx.x; q < nch; q += kThreads)
            reinterpret_cast<uint4*>(lds)[q] = reinterpret_cast<const uint4*>(g)[q];
        for (uint32_t b = (nch << 4) + threadIdx.x; b < nbytes; b += kThreads) lds[b] = g[b];
    } else {
        for (uint32_t b = threadIdx.x; b < nbytes; b += kThreads) lds[b] = g[b];
    }
}

template <int ENV, int DIM>
__device__ __forceinline__ void stage_tables(uint8_t* tab, const uint8_t* par, uint32_t np) {
    using E = EnvT<ENV, DIM>;
    if constexpr (ENV == DCA_ENV_CUBE3) {
        for (uint32_t i = threadIdx.x; i < 12 * 54; i += kThreads) tab[i] = d_cube3_perm.p[i / 54][i % 54];
    } else if constexpr (ENV == DCA_ENV_CUBE4) {
        for (uint32_t i = threadIdx.x; i < 24 * 96; i += kThreads) tab[i] = d_cube4_perm.p[i / 96][i % 96];
    } else if constexpr (ENV == DCA_ENV_LIGHTSOUT) {
        // (no table: the flip mask is arithmetic)
    } else {
        // one lane per parent: locate the blank (n_puzzle.py:51-53) and its 4 swap targets
        for (uint32_t r = threadIdx.x; r < np; r += kThreads) {
            uint32_t z = 0;
            for (int i = E::D - 1; i >= 0; i--)
                if (par[r * E::D + i] == 0) z = (uint32_t)i;  // first zero, like np.where on a valid state
            tab[r * 8] = (uint8_t)z;
            for (int a = 0; a < 4; a++) tab[r * 8 + 1 + a] = (uint8_t)npuzzle_swap(DIM, (int)z, a);
        }
    }
}

// cube3 one-hot rows with 16-bit elements (pytorch_models.py:49-52: index = position * 6 + colour), one 16-byte chunk = 8
// elements.  8 q mod 6 only takes the values 0 / 2 / 4 (q mod 3 = 0 / 1 / 2), so chunk q always starts at column 0, 2 or 4 of
// position P0 = 4 q / 3 and ends inside position P0 + 1: its four words are a window of the SIX words {W(v0,0..2), W(v1,0..2)},
// W(v, k) = the pair of columns 2k, 2k+1 of a position whose colour is v = `one` in the half that is hot, else 0 — a lookup on
// (v >> 1, v & 1) instead of eight compare / select / shift / or chains with a running (column, position, move, parent) counter.
__device__ __forceinline__ void cube3_onehot16_chunk(uint32_t phase, uint32_t v0, uint32_t v1, uint32_t one16, uint32_t (&w)[4]) {
    const uint32_t s0 = (v0 & 1u) ? (one16 << 16) : one16, k0 = v0 >> 1;
    const uint32_t s1 = (v1 & 1u) ? (one16 << 16) : one16, k1 = v1 >> 1;
    const uint32_t a0 = k0 == 0u ? s0 : 0u, a1 = k0 == 1u ? s0 : 0u, a2 = k0 == 2u ? s0 : 0u;
    const uint32_t b0 = k1 == 0u ? s1 : 0u, b1 = k1 == 1u ? s1 : 0u, b2 = k1 == 2u ? s1 : 0u;
    w[0] = phase == 0u ? a0 : phase == 1u ? a1 : a2;
    w[1] = phase == 0u ? a1 : phase == 1u ? a2 : b0;
    w[2] = phase == 0u ? a2 : phase == 1u ? b0 : b1;
    w[3] = phase == 0u ? b0 : phase == 1u ? b1 : b2;
}

// The same for fp32 elements (one 16-byte chunk = 4 elements): 4 q mod 6 only takes 0 / 4 / 2, so chunk q starts at column
// c0 = 0, 2 or 4 (phase = 2 q - 3 P0 = 0, 1, 2; c0 = 2 * phase) of position P0 = 2 q / 3 and its words are
//   c0 = 0: v0 == 0, 1, 2, 3      c0 = 2: v0 == 2, 3, 4, 5      c0 = 4: v0 == 4, 5, then v1 == 0, 1   (v1 = colour of position P0 + 1)
// i.e. words 0, 1 test v0 - c0 against 0 / 1 and words 2, 3 test (phase == 2 ? v1 : v0 - c0 - 2) against 0 / 1.
__device__ __forceinline__ void cube3_onehot32_chunk(uint32_t phase, uint32_t v0, uint32_t v1, uint32_t (&w)[4]) {
    const uint32_t a = v0 - 2u * phase;
    const uint32_t b = phase == 2u ? v1 : a - 2u;
    w[0] = a == 0u ? 0x3F800000u : 0u;
    w[1] = a == 1u ? 0x3F800000u : 0u;
    w[2] = b == 0u ? 0x3F800000u : 0u;
    w[3] = b == 1u ? 0x3F800000u : 0u;
}

// store 16 assembled bytes
__device__ __forceinline__ void store16(uint8_t* dst, const uint32_t (&w)[4], bool aligned) {
    if (aligned) {
        *reinterpret_cast<uint4*>(dst) = make_uint4(w[0], w[1], w[2], w[3]);
    } else {
#pragma unroll
        for (int k = 0; k < 16; k++) dst[k] = (uint8_t)(w[k >> 2] >> (8 * (k & 3)));
    }
}


}  // namespace dca
