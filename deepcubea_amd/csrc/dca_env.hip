// dca_env.hip — cube3 / n-puzzle move, fused expansion, one-hot, is_solved and hash kernels
// for gfx950 (MI355X).  HBM-bound byte work: no MFMA here by design.
//
// Reference behaviour restated (paths relative to forestagostinelli/DeepCubeA):
//   environments/cube3.py:163-171   Cube3._move_np          (permutation gather, 12 moves)
//   environments/cube3.py:129-161   Cube3.expand            (all 12 children per parent)
//   environments/cube3.py:71-85     is_solved, state_to_nnet_input (sticker // 9)
//   environments/n_puzzle.py:216-231 NPuzzle._move_np       (blank swap, 4 moves)
//   utils/pytorch_models.py:49-52   F.one_hot(x.long(), depth).float().view(-1, D*depth)
//
// Kernel shape: one workgroup = 256 threads = a tile of 64 parents.
//   phase 0  coalesced 16-B loads of the parent tile (64*D bytes, contiguous in HBM) into LDS;
//            the move table (cube3: 12x54 gather map; puzzles: per-parent blank + swap slots)
//            is staged in LDS next to it
//   phase 1  one lane per child: 64-bit hash + is_solved straight from LDS gathers
//   phase 2  output-centric streaming: every lane assembles 16 consecutive OUTPUT bytes
//            (children / colour index / one-hot) from LDS gathers and issues one
//            global_store_dwordx4 — every store instruction of a wave covers 1 KiB of
//            contiguous HBM, whatever the 54-byte row stride
// The one-hot stream is 24x (f32) the size of the children and is what bounds the launch.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "dca_common.h"
#include "dca_tile.h"

namespace dca {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int hip_fail(hipError_t e, const char* what) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return (int)e;
}

// --------------------------------------------------------------------------------------------
// fused expansion
// OH: 0 = no one-hot, 4 = f32, 2 = 16-bit (f16/bf16 chosen by `one16`)
// --------------------------------------------------------------------------------------------
template <int ENV, int DIM, int OH>
__global__ __launch_bounds__(kThreads) void expand_fused_kernel(
    const uint8_t* __restrict__ parents, int64_t n, uint8_t* __restrict__ children, uint8_t* __restrict__ nnet_in,
    uint8_t* __restrict__ onehot, uint32_t one16, uint8_t* __restrict__ solved, uint64_t* __restrict__ hash,
    uint32_t align_mask) {
    using E = EnvT<ENV, DIM>;
    using TL = Tile<ENV, DIM>;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* lpar = smem;
    uint8_t* ltab = smem + TL::PAR_BYTES;

    const int64_t p0 = (int64_t)blockIdx.x * kTileParents;
    const uint32_t np = (uint32_t)min((int64_t)kTileParents, n - p0);
    stage_tile(lpar, parents + p0 * E::D, np * E::D, (align_mask & 1) != 0);
    constexpr bool kStaticTable = ENV == DCA_ENV_CUBE3 || ENV == DCA_ENV_CUBE4;  // the move table does not depend on the parents
    if constexpr (kStaticTable) stage_tables<ENV, DIM>(ltab, lpar, np);
    __syncthreads();
    if constexpr (!kStaticTable) {
        stage_tables<ENV, DIM>(ltab, lpar, np);
        __syncthreads();
    }
    TL t{lpar, ltab};
    const uint32_t nchild = np * E::A;
    const int64_t c0 = p0 * E::A;  // first global child index of the tile

    // ---- phase 1: hash + is_solved, one lane per child ----------------------------------
    if (solved != nullptr || hash != nullptr) {
        for (uint32_t c = threadIdx.x; c < nchild; c += kThreads) {
            uint32_t r = c / E::A, a = c - r * E::A;
            uint64_t h = hash_init(E::D);
            bool ok = true;
            uint32_t first = 0;
#pragma unroll
            for (int k = 0; k < E::D; k += 8) {
                uint64_t w = 0;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    if (k + j < E::D) {
                        uint32_t b = t.child_byte(r, a, k + j);
                        ok = solved_step(ENV, E::D, k + j, b, ok, first);
                        w |= (uint64_t)b << (8 * j);
                    }
                }
                h = hash_word(h, w);
            }
            if (hash) hash[c0 + c] = hash_final(h);
            if (solved) solved[c0 + c] = ok ? 1 : 0;
        }
    }

    // ---- phase 2a: children (+ network-input bytes), 16 output bytes per lane -------------
    if (children != nullptr || nnet_in != nullptr) {
        const uint32_t tb = nchild * E::D;
        const int64_t gb = c0 * E::D;
        const uint32_t nch = (tb + 15) >> 4;
        for (uint32_t q = threadIdx.x; q < nch; q += kThreads) {
            uint32_t b0 = q << 4;
            uint32_t c = b0 / E::D, i = b0 - c * E::D;
            uint32_t r = c / E::A, a = c - r * E::A;
            uint32_t w[4] = {0, 0, 0, 0}, v[4] = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 16; k++) {
                uint32_t b = (b0 + k < tb) ? t.child_byte(r, a, i) : 0u;
                w[k >> 2] |= b << (8 * (k & 3));
                if constexpr (ENV == DCA_ENV_CUBE3) v[k >> 2] |= ((b * 57u) >> 9) << (8 * (k & 3));
                if (++i == E::D) {
                    i = 0;
                    if (++a == E::A) {
                        a = 0;
                        ++r;
                    }
                }
            }
            if (b0 + 16 <= tb) {
                if (children) store16(children + gb + b0, w, (align_mask & 2) != 0);
                if (nnet_in) store16(nnet_in + gb + b0, ENV == DCA_ENV_CUBE3 ? v : w, (align_mask & 4) != 0);
            } else {
                for (uint32_t k = 0; b0 + k < tb; k++) {
                    if (children) children[gb + b0 + k] = (uint8_t)(w[k >> 2] >> (8 * (k & 3)));
                    if (nnet_in)
                        nnet_in[gb + b0 + k] = (uint8_t)((ENV == DCA_ENV_CUBE3 ? v : w)[k >> 2] >> (8 * (k & 3)));
                }
            }
        }
    }

    // ---- phase 2b: one-hot rows, 16 output bytes per lane ----------------------------------
    if constexpr (OH != 0) {
        constexpr uint32_t ROW = E::D * E::DEPTH;
        constexpr uint32_t EPC = 16 / OH;  // elements per 16-B chunk
        const uint32_t te = nchild * ROW;  // elements in this tile (< 2^32: 256*2401)
        const int64_t ge = c0 * (int64_t)ROW;
        const uint32_t nch = (te + EPC - 1) / EPC;
        const bool al = (align_mask & 8) != 0;
        for (uint32_t q = threadIdx.x; q < nch; q += kThreads) {
            uint32_t e0 = q * EPC;
            if constexpr (ENV == DCA_ENV_CUBE3 && OH == 2) {
                // 16-bit elements: half the bytes per gathered sticker of the f32 rows, so the generic element loop below (a
                // compare / select / shift / or chain per element) made this variant instruction-bound (0.51 of the HBM peak
                // against 0.66 for f32, round 5).  Two stickers and a lookup per chunk instead (cube3_onehot16_chunk).
                if (e0 + EPC <= te) {
                    const uint32_t P0 = (q << 2) / 3u, phase = (q << 2) - 3u * P0;  // 8 q = 6 P0 + 2 phase
                    const uint32_t c0p = P0 / (uint32_t)E::D, p0 = P0 - c0p * (uint32_t)E::D;
                    const uint32_t r0p = c0p / (uint32_t)E::A, a0p = c0p - r0p * (uint32_t)E::A;
                    uint32_t p1 = p0 + 1u, a1p = a0p, r1p = r0p;  // (one position past the tile on its very last chunk: inside the
                    if (p1 == (uint32_t)E::D) {                    //  LDS allocation, and that chunk only uses W(v1, .) where valid)
                        p1 = 0;
                        if (++a1p == (uint32_t)E::A) {
                            a1p = 0;
                            ++r1p;
                        }
                    }
                    uint32_t w[4];
                    cube3_onehot16_chunk(phase, t.nnet_byte(r0p, a0p, p0), t.nnet_byte(r1p, a1p, p1), one16, w);
                    store16(onehot + (ge + e0) * OH, w, al);
                    continue;
                }
            }
            if constexpr (ENV == DCA_ENV_CUBE3 && OH == 4) {
                // fp32 rows the same way (round 6): one or two stickers and four compares per chunk instead of the element loop's
                // running (column, position, move, parent) counters
                if (e0 + EPC <= te) {
                    const uint32_t P0 = (q << 1) / 3u, phase = (q << 1) - 3u * P0;  // 4 q = 6 P0 + 2 phase
                    const uint32_t c0p = P0 / (uint32_t)E::D, p0 = P0 - c0p * (uint32_t)E::D;
                    const uint32_t r0p = c0p / (uint32_t)E::A, a0p = c0p - r0p * (uint32_t)E::A;
                    uint32_t p1 = p0 + 1u, a1p = a0p, r1p = r0p;
                    if (p1 == (uint32_t)E::D) {
                        p1 = 0;
                        if (++a1p == (uint32_t)E::A) {
                            a1p = 0;
                            ++r1p;
                        }
                    }
                    uint32_t w[4];
                    // (position P0 + 1 is only looked at by the chunks that reach into it; one past the tile: LDS slack, unused)
                    cube3_onehot32_chunk(phase, t.nnet_byte(r0p, a0p, p0), phase == 2u ? t.nnet_byte(r1p, a1p, p1) : 0u, w);
                    store16(onehot + (ge + e0) * OH, w, al);
                    continue;
                }
            }
            uint32_t c = e0 / ROW, e = e0 - c * ROW;
            uint32_t pos = e / E::DEPTH, col = e - pos * E::DEPTH;
            uint32_t r = c / E::A, a = c - r * E::A;
            uint32_t nb = t.nnet_byte(r, a, pos);
            uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
            for (uint32_t k = 0; k < EPC; k++) {
                bool hot = (nb == col) && (e0 + k < te);
                if constexpr (OH == 4) {
                    w[k] = hot ? 0x3F800000u : 0u;
                } else {
                    w[k >> 1] |= (hot ? one16 : 0u) << (16 * (k & 1));
                }
                if (++col == E::DEPTH) {
                    col = 0;
                    if (++pos == E::D) {
                        pos = 0;
                        if (++a == E::A) {
                            a = 0;
                            ++r;
                        }
                    }
                    // r can step to np on the very last element of the tile; the byte read is
                    // still inside the LDS allocation and its value is never used
                    nb = t.nnet_byte(r, a, pos);
                }
            }
            uint8_t* dst = onehot + (ge + e0) * OH;
            if (e0 + EPC <= te) {
                store16(dst, w, al);
            } else {
                for (uint32_t k = 0; e0 + k < te; k++) {
                    if constexpr (OH == 4)
                        reinterpret_cast<uint32_t*>(dst)[k] = w[k];
                    else
                        reinterpret_cast<uint16_t*>(dst)[k] = (uint16_t)(w[k >> 1] >> (16 * (k & 1)));
                }
            }
        }
    }
}

// --------------------------------------------------------------------------------------------
// single-action move (next_state / prev_state): out[n,D]
// --------------------------------------------------------------------------------------------
template <int ENV, int DIM>
__global__ __launch_bounds__(kThreads) void next_state_kernel(const uint8_t* __restrict__ in, int64_t n, int action,
                                                             uint8_t* __restrict__ out, uint32_t align_mask) {
    using E = EnvT<ENV, DIM>;
    using TL = Tile<ENV, DIM>;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* lpar = smem;
    uint8_t* ltab = smem + TL::PAR_BYTES;
    const int64_t p0 = (int64_t)blockIdx.x * kTileParents;
    const uint32_t np = (uint32_t)min((int64_t)kTileParents, n - p0);
    stage_tile(lpar, in + p0 * E::D, np * E::D, (align_mask & 1) != 0);
    constexpr bool kStaticTable = ENV == DCA_ENV_CUBE3 || ENV == DCA_ENV_CUBE4;
    if constexpr (kStaticTable) stage_tables<ENV, DIM>(ltab, lpar, np);
    __syncthreads();
    if constexpr (!kStaticTable) {
        stage_tables<ENV, DIM>(ltab, lpar, np);
        __syncthreads();
    }
    TL t{lpar, ltab};
    const uint32_t tb = np * E::D;
    const int64_t gb = p0 * E::D;
    const uint32_t nch = (tb + 15) >> 4;
    const uint32_t a = (uint32_t)action;
    for (uint32_t q = threadIdx.x; q < nch; q += kThreads) {
        uint32_t b0 = q << 4;
        uint32_t r = b0 / E::D, i = b0 - r * E::D;
        uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 16; k++) {
            uint32_t b = (b0 + k < tb) ? t.child_byte(r, a, i) : 0u;
            w[k >> 2] |= b << (8 * (k & 3));
            if (++i == E::D) {
                i = 0;
                ++r;
            }
        }
        if (b0 + 16 <= tb)
            store16(out + gb + b0, w, (align_mask & 2) != 0);
        else
            for (uint32_t k = 0; b0 + k < tb; k++) out[gb + b0 + k] = (uint8_t)(w[k >> 2] >> (8 * (k & 3)));
    }
}

// --------------------------------------------------------------------------------------------
// stand-alone per-state kernels (Environment mirror; not on the throughput path)
// --------------------------------------------------------------------------------------------
__global__ void state_scan_kernel(int env, const uint8_t* __restrict__ st, int64_t n, int D, uint8_t* solved,
                                  uint64_t* hash, float* heur, int heur_id) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t* s = st + i * D;
    uint64_t h = hash_init(D), sum = 0;
    uint32_t manh = 0;
    const int pdim = D == 16 ? 4 : D == 25 ? 5 : D == 36 ? 6 : D == 49 ? 7 : 0;  // sliding-puzzle side (0: cube3)
    bool ok = true;
    uint32_t first = 0;
    for (int k = 0; k < D; k += 8) {
        uint64_t w = 0;
        for (int j = 0; j < 8 && k + j < D; j++) {
            uint32_t b = s[k + j];
            ok = solved_step(env, D, k + j, b, ok, first);
            w |= (uint64_t)b << (8 * j);
            sum += (uint64_t)b * (uint64_t)(7 * (k + j) + 3);
            if (pdim) manh += manhattan_term(pdim, (uint32_t)(k + j), b);
        }
        h = hash_word(h, w);
    }
    h = hash_final(h);
    if (solved) solved[i] = ok ? 1 : 0;
    if (hash) hash[i] = h;
    if (heur) heur[i] = heur_from(heur_id, sum, h, manh);
}

__global__ void nnet_input_kernel(int env, const uint8_t* __restrict__ st, int64_t nbytes, uint8_t* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < nbytes; i += stride) {
        uint32_t b = st[i];
        out[i] = (uint8_t)(env == DCA_ENV_CUBE3 ? (b * 57u) >> 9 : b);
    }
}

template <int OH>
__global__ void onehot_kernel(const uint8_t* __restrict__ idx, int64_t n, int D, int depth, uint8_t* __restrict__ out,
                              uint32_t one16) {
    // one lane per output element group of EPC; generic (any D, depth) — used by the host mirror
    const int64_t total = n * D * depth;
    int64_t e = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x);
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; e < total; e += stride) {
        int64_t p = e / depth;
        uint32_t col = (uint32_t)(e - p * depth);
        bool hot = idx[p] == col;
        if constexpr (OH == 4)
            reinterpret_cast<uint32_t*>(out)[e] = hot ? 0x3F800000u : 0u;
        else
            reinterpret_cast<uint16_t*>(out)[e] = hot ? (uint16_t)one16 : (uint16_t)0;
    }
}

static inline uint32_t one16_of(int dtype) { return dtype == DCA_DT_F16 ? 0x3C00u : 0x3F80u; }
static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <int ENV, int DIM>
static int launch_expand(const uint8_t* parents, int64_t n, uint8_t* children, uint8_t* nnet_in, void* onehot,
                         int onehot_dtype, uint8_t* solved, uint64_t* hash, hipStream_t s) {
    using TL = Tile<ENV, DIM>;
    if (n == 0) return 0;
    int64_t blocks = (n + kTileParents - 1) / kTileParents;
    DCA_ARG(blocks < (1ll << 31));
    uint32_t am = (al16(parents) ? 1u : 0u) | (al16(children) ? 2u : 0u) | (al16(nnet_in) ? 4u : 0u) |
                  (al16(onehot) ? 8u : 0u);
    dim3 g((unsigned)blocks), b(kThreads);
    size_t lds = TL::LDS_BYTES;
    if (onehot == nullptr) {
        hipLaunchKernelGGL((expand_fused_kernel<ENV, DIM, 0>), g, b, lds, s, parents, n, children, nnet_in, nullptr,
                           0u, solved, hash, am);
    } else if (onehot_dtype == DCA_DT_F32) {
        hipLaunchKernelGGL((expand_fused_kernel<ENV, DIM, 4>), g, b, lds, s, parents, n, children, nnet_in,
                           (uint8_t*)onehot, 0u, solved, hash, am);
    } else {
        hipLaunchKernelGGL((expand_fused_kernel<ENV, DIM, 2>), g, b, lds, s, parents, n, children, nnet_in,
                           (uint8_t*)onehot, one16_of(onehot_dtype), solved, hash, am);
    }
    return launch_check("expand_fused_kernel");
}

template <int ENV, int DIM>
static int launch_next(const uint8_t* in, int64_t n, int action, uint8_t* out, hipStream_t s) {
    using TL = Tile<ENV, DIM>;
    if (n == 0) return 0;
    int64_t blocks = (n + kTileParents - 1) / kTileParents;
    DCA_ARG(blocks < (1ll << 31));
    uint32_t am = (al16(in) ? 1u : 0u) | (al16(out) ? 2u : 0u);
    hipLaunchKernelGGL((next_state_kernel<ENV, DIM>), dim3((unsigned)blocks), dim3(kThreads), TL::LDS_BYTES, s, in, n,
                       action, out, am);
    return launch_check("next_state_kernel");
}

// Store-only yardstick for the gather kernel (dca_debug_write_ceiling; the stand-alone tools/hbm_write_ceiling.hip found this
// pattern the best on the chip): every workgroup fills its own contiguous region with plain 16-byte stores, 1 KiB per wave
// instruction — what expand_fused_kernel's output phase does, minus everything that produces the bytes.
__global__ __launch_bounds__(kThreads) void k_write_ceiling(uint4* __restrict__ p, uint64_t chunks_per_block, uint64_t chunks) {
    const uint64_t c0 = (uint64_t)blockIdx.x * chunks_per_block;
    const uint64_t c1 = c0 + chunks_per_block < chunks ? c0 + chunks_per_block : chunks;
    const uint4 v = make_uint4(blockIdx.x, 1u, 2u, 3u);
    for (uint64_t i = c0 + threadIdx.x; i < c1; i += kThreads) p[i] = v;
}

// internal entry points used by the engine (dca_engine.hip)
int expand_dispatch(int env, int dim, const uint8_t* parents, int64_t n, uint8_t* children, uint8_t* nnet_in,
                    void* onehot, int onehot_dtype, uint8_t* solved, uint64_t* hash, hipStream_t s) {
    if (env == DCA_ENV_CUBE3)
        return launch_expand<DCA_ENV_CUBE3, 0>(parents, n, children, nnet_in, onehot, onehot_dtype, solved, hash, s);
    if (env == DCA_ENV_CUBE4) return launch_expand<DCA_ENV_CUBE4, 0>(parents, n, children, nullptr, nullptr, 0, solved, hash, s);
    if (env == DCA_ENV_LIGHTSOUT) {
        if (dim == 7) return launch_expand<DCA_ENV_LIGHTSOUT, 7>(parents, n, children, nnet_in, onehot, onehot_dtype, solved, hash, s);
        set_error("unsupported lightsout dim %d (7)", dim);
        return DCA_E_BADARG;
    }
    switch (dim) {
        case 4: return launch_expand<DCA_ENV_NPUZZLE, 4>(parents, n, children, nnet_in, onehot, onehot_dtype, solved, hash, s);
        case 5: return launch_expand<DCA_ENV_NPUZZLE, 5>(parents, n, children, nnet_in, onehot, onehot_dtype, solved, hash, s);
        case 6: return launch_expand<DCA_ENV_NPUZZLE, 6>(parents, n, children, nnet_in, onehot, onehot_dtype, solved, hash, s);
        case 7: return launch_expand<DCA_ENV_NPUZZLE, 7>(parents, n, children, nnet_in, onehot, onehot_dtype, solved, hash, s);
    }
    set_error("unsupported puzzle dim %d (4..7)", dim);
    return DCA_E_BADARG;
}

}  // namespace dca

using namespace dca;

extern "C" {

int dca_abi_version(void) { return DCA_ABI_VERSION; }
const char* dca_last_error(void) { return g_err; }

const uint8_t* dca_cube3_perm_table(void) { return &kCube3Perm.p[0][0]; }

int dca_npuzzle_swap_table(int dim, uint8_t* out) {
    DCA_ARG(dim >= 4 && dim <= 7 && out != nullptr);
    for (int z = 0; z < dim * dim; z++)
        for (int a = 0; a < 4; a++) out[z * 4 + a] = (uint8_t)npuzzle_swap(dim, z, a);
    return 0;
}

int dca_cube3_next_state(const uint8_t* in, int64_t n, int action, uint8_t* out, void* stream) {
    DCA_ARG(n >= 0 && action >= 0 && action < 12 && (n == 0 || (in && out)));
    return launch_next<DCA_ENV_CUBE3, 0>(in, n, action, out, (hipStream_t)stream);
}
int dca_cube3_prev_state(const uint8_t* in, int64_t n, int action, uint8_t* out, void* stream) {
    // moves_rev (cube3.py:29) pairs move a with a^1
    DCA_ARG(action >= 0 && action < 12);
    return dca_cube3_next_state(in, n, action ^ 1, out, stream);
}
int dca_npuzzle_next_state(const uint8_t* in, int64_t n, int dim, int action, uint8_t* out, void* stream) {
    DCA_ARG(n >= 0 && action >= 0 && action < 4 && (n == 0 || (in && out)));
    hipStream_t s = (hipStream_t)stream;
    switch (dim) {
        case 4: return launch_next<DCA_ENV_NPUZZLE, 4>(in, n, action, out, s);
        case 5: return launch_next<DCA_ENV_NPUZZLE, 5>(in, n, action, out, s);
        case 6: return launch_next<DCA_ENV_NPUZZLE, 6>(in, n, action, out, s);
        case 7: return launch_next<DCA_ENV_NPUZZLE, 7>(in, n, action, out, s);
    }
    set_error("unsupported puzzle dim %d (4..7)", dim);
    return DCA_E_BADARG;
}
int dca_npuzzle_prev_state(const uint8_t* in, int64_t n, int dim, int action, uint8_t* out, void* stream) {
    DCA_ARG(action >= 0 && action < 4);  // moves_rev = D,U,R,L (n_puzzle.py:29)
    return dca_npuzzle_next_state(in, n, dim, action ^ 1, out, stream);
}

int dca_debug_write_ceiling(void* buf, int64_t bytes, int64_t bytes_per_block, void* stream) {
    DCA_ARG(buf != nullptr && bytes >= 16 && bytes_per_block >= 16 && (reinterpret_cast<uintptr_t>(buf) & 15) == 0);
    const uint64_t chunks = (uint64_t)bytes / 16, cpb = (uint64_t)bytes_per_block / 16;
    const uint64_t blocks = (chunks + cpb - 1) / cpb;
    DCA_ARG(blocks < (1ull << 31));
    hipLaunchKernelGGL(k_write_ceiling, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream, (uint4*)buf, cpb, chunks);
    return launch_check("k_write_ceiling");
}

const uint8_t* dca_cube4_perm_table(void) { return &kCube4Perm.p[0][0]; }

int dca_cube4_next_state(const uint8_t* in, int64_t n, int action, uint8_t* out, void* stream) {
    DCA_ARG(n >= 0 && action >= 0 && action < 24 && (n == 0 || (in && out)));
    return launch_next<DCA_ENV_CUBE4, 0>(in, n, action, out, (hipStream_t)stream);
}
int dca_cube4_prev_state(const uint8_t* in, int64_t n, int action, uint8_t* out, void* stream) {
    DCA_ARG(action >= 0 && action < 24);  // the "_1" table of a move is the inverse of its "_n1" table (cpp:264-320)
    return dca_cube4_next_state(in, n, action ^ 1, out, stream);
}
int dca_cube4_expand_fused(const uint8_t* parents, int64_t n, uint8_t* children, uint8_t* is_solved, uint64_t* hash, void* stream) {
    DCA_ARG(n >= 0 && (n == 0 || parents));
    return expand_dispatch(DCA_ENV_CUBE4, 0, parents, n, children, nullptr, nullptr, 0, is_solved, hash, (hipStream_t)stream);
}

int dca_lightsout_next_state(const uint8_t* in, int64_t n, int dim, int action, uint8_t* out, void* stream) {
    DCA_ARG(dim == 7 && n >= 0 && action >= 0 && action < dim * dim && (n == 0 || (in && out)));
    return launch_next<DCA_ENV_LIGHTSOUT, 7>(in, n, action, out, (hipStream_t)stream);
}

int dca_lightsout_expand_fused(const uint8_t* parents, int64_t n, int dim, uint8_t* children, void* onehot, int onehot_dtype,
                               uint8_t* is_solved, uint64_t* hash, void* stream) {
    DCA_ARG(n >= 0 && (n == 0 || parents));
    DCA_ARG(onehot == nullptr || (onehot_dtype >= DCA_DT_F32 && onehot_dtype <= DCA_DT_BF16));
    return expand_dispatch(DCA_ENV_LIGHTSOUT, dim, parents, n, children, nullptr, onehot, onehot_dtype, is_solved, hash,
                           (hipStream_t)stream);
}

int dca_cube3_expand_fused(const uint8_t* parents, int64_t n, uint8_t* children, uint8_t* color_idx, void* onehot,
                           int onehot_dtype, uint8_t* is_solved, uint64_t* hash, void* stream) {
    DCA_ARG(n >= 0 && (n == 0 || parents));
    DCA_ARG(onehot == nullptr || (onehot_dtype >= DCA_DT_F32 && onehot_dtype <= DCA_DT_BF16));
    return expand_dispatch(DCA_ENV_CUBE3, 0, parents, n, children, color_idx, onehot, onehot_dtype, is_solved, hash,
                           (hipStream_t)stream);
}
int dca_npuzzle_expand_fused(const uint8_t* parents, int64_t n, int dim, uint8_t* children, void* onehot,
                             int onehot_dtype, uint8_t* is_solved, uint64_t* hash, void* stream) {
    DCA_ARG(n >= 0 && (n == 0 || parents));
    DCA_ARG(onehot == nullptr || (onehot_dtype >= DCA_DT_F32 && onehot_dtype <= DCA_DT_BF16));
    return expand_dispatch(DCA_ENV_NPUZZLE, dim, parents, n, children, nullptr, onehot, onehot_dtype, is_solved, hash,
                           (hipStream_t)stream);
}

static int state_dim_of(int env, int dim, int* D) {
    if (env == DCA_ENV_CUBE3) {
        *D = 54;
        return 0;
    }
    if (env == DCA_ENV_CUBE4) {
        *D = 96;
        return 0;
    }
    if ((env == DCA_ENV_NPUZZLE && dim >= 4 && dim <= 7) || (env == DCA_ENV_LIGHTSOUT && dim == 7)) {
        *D = dim * dim;
        return 0;
    }
    set_error("unknown env %d / dim %d", env, dim);
    return DCA_E_BADARG;
}

int dca_is_solved(int env, int dim, const uint8_t* states, int64_t n, uint8_t* out, void* stream) {
    int D;
    if (int rc = state_dim_of(env, dim, &D)) return rc;
    DCA_ARG(n >= 0 && (n == 0 || (states && out)));
    if (n == 0) return 0;
    hipLaunchKernelGGL(state_scan_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, env,
                       states, n, D, out, nullptr, nullptr, 0);
    return launch_check("state_scan_kernel");
}
int dca_hash64(const uint8_t* states, int64_t n, int state_dim, uint64_t* out, void* stream) {
    DCA_ARG(n >= 0 && state_dim > 0 && state_dim <= 96 && (n == 0 || (states && out)));
    if (n == 0) return 0;
    hipLaunchKernelGGL(state_scan_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       DCA_ENV_CUBE3, states, n, state_dim, nullptr, out, nullptr, 0);
    return launch_check("state_scan_kernel");
}
int dca_heuristic_builtin(int heur_id, const uint8_t* states, int64_t n, int state_dim, float* out, void* stream) {
    DCA_ARG(heur_id >= 0 && heur_id <= DCA_HEUR_MANHATTAN && n >= 0 && state_dim > 0 && state_dim <= 64);
    DCA_ARG(n == 0 || (states && out));
    if (n == 0) return 0;
    hipLaunchKernelGGL(state_scan_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       DCA_ENV_CUBE3, states, n, state_dim, nullptr, nullptr, out, heur_id);
    return launch_check("state_scan_kernel");
}
int dca_nnet_input(int env, int dim, const uint8_t* states, int64_t n, uint8_t* out, void* stream) {
    int D;
    if (int rc = state_dim_of(env, dim, &D)) return rc;
    DCA_ARG(n >= 0 && (n == 0 || (states && out)));
    if (n == 0) return 0;
    int64_t nb = n * D;
    unsigned blocks = (unsigned)((nb + 255) / 256 < 4096 ? (nb + 255) / 256 : 4096);
    hipLaunchKernelGGL(nnet_input_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, env, states, nb, out);
    return launch_check("nnet_input_kernel");
}
int dca_onehot(const uint8_t* idx, int64_t n, int state_dim, int depth, void* out, int dtype, void* stream) {
    DCA_ARG(n >= 0 && state_dim > 0 && depth > 0 && depth <= 256 && dtype >= DCA_DT_F32 && dtype <= DCA_DT_BF16);
    DCA_ARG(n == 0 || (idx && out));
    if (n == 0) return 0;
    int64_t total = n * state_dim * depth;
    unsigned blocks = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (dtype == DCA_DT_F32)
        hipLaunchKernelGGL(onehot_kernel<4>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, idx, n, state_dim, depth,
                           (uint8_t*)out, 0u);
    else
        hipLaunchKernelGGL(onehot_kernel<2>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, idx, n, state_dim, depth,
                           (uint8_t*)out, one16_of(dtype));
    return launch_check("onehot_kernel");
}

}  // extern "C"
