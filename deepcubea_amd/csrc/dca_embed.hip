// dca_embed.hip — first layer of the cost-to-go network as an EMBEDDING SUM ("embedding-sum layer 1 fed directly
// by the expand kernel (no one-hot materialisation)").
//
// Reference arithmetic (utils/pytorch_models.py:49-60, BatchNorm folded): y = relu(onehot(s) . W1^T + b1).  A one-hot row has
// exactly D ones — one per position — so
//            y[r, n] = relu( b1[n] + sum_pos W1[n, pos * DEPTH + s[r, pos]] ):
// D gathered weights per output instead of D * DEPTH multiply-adds.  On the matrix pipes (csrc/dca_mlp.hip) the one-hot GEMM
// pays for all D * DEPTH columns: fine for cube3 (DEPTH = 6: 324 columns, and the MFMA rate is 16x the VALU's), ruinous for the
// sliding puzzles whose DEPTH is the tile count — puzzle48 (BASELINE configs[4]): 2401 columns for 49 ones, layer 1 = half of the
// network's flops, 29 ms of the parity mode's 56 ms per 409 600 rows.  As a sum of 49 gathered rows it is 1/49 of the work, in
// EXACT fp32 arithmetic (no operand splitting), and LDS-bandwidth bound:
//
//   * a workgroup (8 waves) owns NT output columns; their fp32 weights, transposed — [K][NT], one 4 * NT-byte row per one-hot
//     column — sit in LDS for the workgroup's whole life (NT = 64 for cube3: 83 KB; 16 for puzzle48: 154 KB);
//   * it walks over chunks of R states: the chunk's R * D state bytes come in by 16-byte loads (the next chunk's are prefetched
//     into registers under the current one's sums) into a linear LDS image; a lane owns (state, 4 consecutive columns): it pulls its
//     state's bytes out of the image as aligned dwords + v_alignbyte, then per position one v_bfe, one v_mad (row address), one
//     ds_read_b128 and four adds — positions in ascending order, so a state's value has the same bits in any batch;
//   * the NT / 4 lanes of a state read 4 * NT contiguous bytes of one LDS row: the lane groups of a ds_read_b128 touch disjoint
//     bank ranges for NT = 64, and the tail stores 2 * NT contiguous bytes per state and plane (fp16 planes for the f16x3 layers,
//     bf16, fp32, or e4m3 bytes for the fp8 layers).
#include "dca_common.h"

namespace dca {

constexpr int kEmbThreads = 512;

template <int D, int DEPTH, int NT, int R>
struct EmbGeo {
    static constexpr int K = D * DEPTH;
    static constexpr int W_BYTES = K * NT * 4;
    static constexpr int ST_BYTES = ((R * D + 15) / 16) * 16 + 16;  // the chunk's state bytes (+ slack: the dword reads run past the last row)
    static constexpr int LDS = W_BYTES + NT * 4 + ST_BYTES;
    static constexpr int NW = (D + 3) / 4 + 1;                       // aligned dwords covering one state row at any byte offset
    static constexpr int NPRE = (R * D + 16 * kEmbThreads - 1) / (16 * kEmbThreads);  // 16-byte pieces per thread of a chunk's states
};

template <int D, int DEPTH, int NT, int R, int OUT /*0 fp32, 2 bf16, 4 two fp16 planes (high, then low at + m * ldo), 5 e4m3 (saturating)*/>
__global__ __launch_bounds__(kEmbThreads) void k_l1_embed(const uint8_t* __restrict__ nn, int64_t m, const float* __restrict__ wt /*[K][n_pad]*/,
                                                          int64_t n_pad, const float* __restrict__ bias, int relu, void* __restrict__ out,
                                                          int* __restrict__ overflow) {
    using G = EmbGeo<D, DEPTH, NT, R>;
    extern __shared__ __attribute__((aligned(16))) uint8_t le[];
    float* lw = reinterpret_cast<float*>(le);
    float* lb = lw + G::K * NT;
    uint8_t* ls = le + G::W_BYTES + NT * 4;
    const int t = threadIdx.x;
    const int64_t n0 = (int64_t)blockIdx.x * NT;
    const int64_t nchunks = (m + R - 1) / R;

    uint4 pre[G::NPRE];
    auto prefetch = [&](int64_t chunk) {  // chunk's state bytes -> registers (zeros past the matrix: a zero byte is a valid colour)
        const int64_t b0 = chunk * (int64_t)R * D, nb = m * D - b0;  // bytes left from the chunk's start (<= 0: nothing)
#pragma unroll
        for (int j = 0; j < G::NPRE; j++) {
            const int64_t q = (int64_t)(t + j * kEmbThreads) * 16;
            pre[j] = make_uint4(0u, 0u, 0u, 0u);
            if (q + 16 <= nb && q < (int64_t)R * D) {
                pre[j] = *reinterpret_cast<const uint4*>(nn + b0 + q);
            } else if (q < nb && q < (int64_t)R * D) {  // the matrix's last, partial piece: byte by byte (nothing is read past its end)
                uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                for (int e = 0; e < 16; e++)
                    if (q + e < nb) w[e >> 2] |= (uint32_t)nn[b0 + q + e] << (8 * (e & 3));
                pre[j] = make_uint4(w[0], w[1], w[2], w[3]);
            }
        }
    };
    prefetch(blockIdx.y);
    // the tile's weights (row k = one-hot column k: NT floats of the transposed matrix) and its bias
    for (int q = t; q < G::K * (NT / 4); q += kEmbThreads) {
        const int k = q / (NT / 4), c = q - k * (NT / 4);
        reinterpret_cast<float4*>(lw)[q] = *reinterpret_cast<const float4*>(wt + (int64_t)k * n_pad + n0 + 4 * c);
    }
    if (t < NT) lb[t] = bias[n0 + t];
    bool ovf = false;
    for (int64_t chunk = blockIdx.y; chunk < nchunks; chunk += gridDim.y) {
        __syncthreads();  // the previous chunk's readers are done with the state image (first pass: the weights are staged)
#pragma unroll
        for (int j = 0; j < G::NPRE; j++) {
            const int q = (t + j * kEmbThreads) * 16;
            if (q < G::ST_BYTES - 16) *reinterpret_cast<uint4*>(ls + q) = pre[j];
        }
        __syncthreads();
        prefetch(chunk + gridDim.y);  // flies under this chunk's sums
        const int64_t r0 = chunk * R;
        for (int task = t; task < R * (NT / 4); task += kEmbThreads) {
            const int st = task / (NT / 4), cp = task - st * (NT / 4);
            // the state's D bytes from the linear image: aligned dwords, shifted into place
            const uint32_t boff = (uint32_t)st * D, sh = boff & 3u;
            const uint32_t* wsrc = reinterpret_cast<const uint32_t*>(ls + (boff & ~3u));
            uint32_t w[G::NW];
#pragma unroll
            for (int i = 0; i < G::NW; i++) w[i] = wsrc[i];
            uint32_t v[G::NW - 1];
#pragma unroll
            for (int i = 0; i < G::NW - 1; i++) v[i] = __builtin_amdgcn_alignbyte(w[i + 1], w[i], sh);
            float4 acc = *reinterpret_cast<const float4*>(lb + 4 * cp);
            const uint8_t* wcol = reinterpret_cast<const uint8_t*>(lw) + cp * 16;
#pragma unroll
            for (int pos = 0; pos < D; pos++) {
                const uint32_t s = (v[pos >> 2] >> (8 * (pos & 3))) & 0xFFu;
                const float4 g = *reinterpret_cast<const float4*>(wcol + (s + (uint32_t)(pos * DEPTH)) * (uint32_t)(NT * 4));
                acc.x += g.x;
                acc.y += g.y;
                acc.z += g.z;
                acc.w += g.w;
            }
            float u[4] = {acc.x, acc.y, acc.z, acc.w};
            if (relu) {
#pragma unroll
                for (int e = 0; e < 4; e++) u[e] = fmaxf(u[e], 0.f);
            }
            const int64_t r = r0 + st;
            if (r < m) {
                const int64_t o = r * n_pad + n0 + 4 * cp;
                if constexpr (OUT == 0) {
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + o) = make_float4(u[0], u[1], u[2], u[3]);
                } else if constexpr (OUT == 2) {
                    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
                    typedef float f2 __attribute__((ext_vector_type(2)));
                    const f2 a = {u[0], u[1]}, b = {u[2], u[3]};
                    const bf2 pa = __builtin_convertvector(a, bf2), pb = __builtin_convertvector(b, bf2);
                    uint2 q;
                    __builtin_memcpy(&q.x, &pa, 4);
                    __builtin_memcpy(&q.y, &pb, 4);
                    *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(out) + o) = q;
                } else if constexpr (OUT == 5) {  // e4m3fn has no infinity: saturate at +-448
                    auto sat = [](float f) { return fminf(fmaxf(f, -448.f), 448.f); };
                    uint32_t q = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(sat(u[0]), sat(u[1]), 0, false);
                    q = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(sat(u[2]), sat(u[3]), (int)q, true);
                    *reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(out) + o) = q;
                } else {
                    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                    h4 hi, lo;
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        ovf |= !(fabsf(u[e]) <= 60000.0f);
                        hi[e] = (_Float16)u[e];
                        lo[e] = (_Float16)(u[e] - (float)hi[e]);
                    }
                    _Float16* q = reinterpret_cast<_Float16*>(out) + o;
                    *reinterpret_cast<h4*>(q) = hi;
                    *reinterpret_cast<h4*>(q + m * n_pad) = lo;
                }
            }
        }
    }
    if (OUT == 4 && ovf && overflow) *overflow = 1;
}

template <int D, int DEPTH, int NT, int R>
int launch_embed(const uint8_t* nn, int64_t m, const float* wt, int64_t n_pad, const float* bias, int relu, void* out, int out_dtype,
                 int* overflow, hipStream_t s) {
    using G = EmbGeo<D, DEPTH, NT, R>;
    static_assert(G::LDS <= 160 * 1024, "weight slice does not fit LDS");
    static_assert(NT % 4 == 0 && (R * D) % 16 == 0, "tile geometry");
    if (n_pad % NT != 0) {
        set_error("dca_l1_embed: n_pad %lld is not a multiple of the column tile %d", (long long)n_pad, NT);
        return DCA_E_BADARG;
    }
    const int64_t chunks = (m + R - 1) / R, tiles = n_pad / NT;
    int64_t gy = (1024 + tiles - 1) / tiles;  // ~4 workgroups per CU over the launch (one resident per CU at a time)
    if (gy > chunks) gy = chunks;
    if (gy < 1) gy = 1;
    const dim3 grid((unsigned)tiles, (unsigned)gy), block(kEmbThreads);
#define DCA_EMB_LAUNCH(OUTV)                                                                                            \
    do {                                                                                                                \
        auto kern = k_l1_embed<D, DEPTH, NT, R, OUTV>;                                                                  \
        DCA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS)); \
        hipLaunchKernelGGL(kern, grid, block, G::LDS, s, nn, m, wt, n_pad, bias, relu, out, overflow);                  \
    } while (0)
    if (out_dtype == DCA_DT_F32)
        DCA_EMB_LAUNCH(0);
    else if (out_dtype == DCA_DT_BF16)
        DCA_EMB_LAUNCH(2);
    else if (out_dtype == DCA_DT_E4M3)
        DCA_EMB_LAUNCH(5);
    else
        DCA_EMB_LAUNCH(4);
#undef DCA_EMB_LAUNCH
    return launch_check("k_l1_embed");
}

}  // namespace dca

using namespace dca;

extern "C" {

int dca_l1_embed_supported(int state_dim, int depth) {
    return (state_dim == 54 && depth == 6) || (state_dim == depth && (depth == 16 || depth == 25 || depth == 36 || depth == 49)) ||
           (state_dim == 49 && depth == 6);
}

int dca_l1_embed(const uint8_t* nnet_in, int64_t m, int state_dim, int depth, const float* w_t, int64_t n_pad, const float* bias,
                 int relu, void* out, int out_dtype, int* overflow, void* stream) {
    DCA_ARG(nnet_in && w_t && bias && out && m >= 0 && n_pad >= 64 && n_pad % 64 == 0);
    DCA_ARG(out_dtype == DCA_DT_F32 || out_dtype == DCA_DT_BF16 || out_dtype == DCA_DT_F16_PLANES || out_dtype == DCA_DT_E4M3);
    DCA_ARG(((reinterpret_cast<uintptr_t>(nnet_in) | reinterpret_cast<uintptr_t>(w_t) | reinterpret_cast<uintptr_t>(out)) & 15) == 0);
    if (!dca_l1_embed_supported(state_dim, depth)) {
        set_error("dca_l1_embed: geometry (%d, %d) not instantiated", state_dim, depth);
        return DCA_E_BADARG;
    }
    if (m == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (state_dim == 54) return launch_embed<54, 6, 64, 512>(nnet_in, m, w_t, n_pad, bias, relu, out, out_dtype, overflow, s);
    if (state_dim == 16) return launch_embed<16, 16, 64, 512>(nnet_in, m, w_t, n_pad, bias, relu, out, out_dtype, overflow, s);
    if (state_dim == 25) return launch_embed<25, 25, 32, 512>(nnet_in, m, w_t, n_pad, bias, relu, out, out_dtype, overflow, s);
    if (state_dim == 36) return launch_embed<36, 36, 16, 512>(nnet_in, m, w_t, n_pad, bias, relu, out, out_dtype, overflow, s);
    if (depth == 49) return launch_embed<49, 49, 16, 128>(nnet_in, m, w_t, n_pad, bias, relu, out, out_dtype, overflow, s);
    return launch_embed<49, 6, 64, 512>(nnet_in, m, w_t, n_pad, bias, relu, out, out_dtype, overflow, s);
}

}  // extern "C"
