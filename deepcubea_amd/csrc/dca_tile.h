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
