// dca_mlp8.hip — first layer of the cost-to-go network in the NON-parity fp8 mode (`--nnet_dtype fp8`) on gfx950's f8f6f4 matrix
// pipe.
//
// Reference arithmetic (utils/pytorch_models.py:49-60, BatchNorm folded): y = relu(onehot(s) . W1^T + b1).  A one-hot row is
// EXACT in OCP e4m3 (its entries are 0 and 1.0 = 0x38), and every other layer of the fp8 mode already keeps its weights as
// e4m3 bytes with one fp32 scale per output unit — so layer 1 can run on v_mfma_f32_32x32x64_f8f6f4 (twice the bf16 rate:
// K = 324 -> 384 is 6 instructions of 64 cycles per 32 x 32 block instead of 21 of 32) instead of multiplying exact 0 / 1 rows
// on the bf16 pipe and only ROUNDING to e4m3 at the end, which is what dca_l1_onehot_gemm(..., DCA_DT_E4M3) does (1.0 ms per
// 204 800 x 5120 layer, 20 % of the fp8 forward in round 5).
//
//   out8[r, n] = e4m3(sat( relu?( (onehot(s_r) . w8[n, :]) * scale[n] + bias[n] ) ))        (the caller folds the activation scale
//                                                                                           of the layer's output into scale / bias)
//
// Structure (one workgroup = 8 waves = 512 state rows x 128 output columns, persistent over row chunks):
//   * the tile's weights — 128 columns x KPAD bytes, 48 KB for cube3 — are staged ONCE per workgroup into LDS in fragment order
//     [k / 16][column][16]: a wave's ds_read_b128 of a 16-byte K piece covers 32 consecutive columns = 512 contiguous bytes;
//   * the MFMA's operand roles are the lean ones of csrc/dca_mlp.hip: weights = A operand (D rows = output columns), one-hot
//     rows = B operand (D columns = state rows), weight column of MFMA row i chosen with bits 2 and 3 of i swapped, so that a
//     lane's 8 consecutive accumulator registers are 8 consecutive output columns of ONE state row;
//   * a wave's 64 state rows (64 x D contiguous bytes, 16-byte aligned) come in by 16-byte loads through a wave-private LDS slice;
//     lane (j, h) builds the K-bit one-hot mask of ONE row (row j for h = 0, row 32 + j for h = 1) and one v_permlane32_swap
//     per K step hands each lane the 32 mask bits it feeds the instruction with for both of its rows (word 2 s + h of each);
//     a mask byte becomes eight e4m3 bytes through a 256-entry table in LDS; the rows of the NEXT chunk are prefetched into
//     registers under the K loop;
//   * tail: scale / bias from LDS, ReLU + saturation as one v_med3, v_cvt_pk_fp8_f32, 8-byte pieces exchanged inside the wave
//     through 4 KB of LDS so that 8 lanes store one 128-byte row segment (full lines).
#include "dca_common.h"

namespace dca {

typedef int l8_i32x4 __attribute__((ext_vector_type(4)));
typedef int l8_i32x8 __attribute__((ext_vector_type(8)));
typedef float l8_f32x16 __attribute__((ext_vector_type(16)));

constexpr int kL8Threads = 512;
constexpr int kL8Cols = 128;

template <int D, int DEPTH>
struct L8Geo {
    static constexpr int K = D * DEPTH;
    static constexpr int KSTEPS = (K + 63) / 64;
    static constexpr int KPAD = KSTEPS * 64;
    static constexpr int KC16 = KPAD / 16;
    static constexpr int MW = KPAD / 32;  // one-hot mask words per row
    static constexpr int W_BYTES = KPAD * kL8Cols;
    static constexpr int ROW_BYTES = ((64 * D + 15) / 16) * 16 + 16;  // a wave's 64 rows (+ slack for the last 16-byte piece)
    static constexpr int MISC_BYTES = 2 * kL8Cols * 4 + 256 * 8;  // scale, bias, the byte -> eight e4m3 bytes table
    static constexpr int NPRE = (64 * D + 1023) / 1024;              // 16-byte row pieces per lane of a wave's 64 rows
    static constexpr int LDS = W_BYTES + MISC_BYTES + (kL8Threads / 64) * (4096 + ROW_BYTES);
};

template <int D, int DEPTH>
__global__ __launch_bounds__(kL8Threads) void k_l1_onehot_gemm8(const uint8_t* __restrict__ nn, int64_t m,
                                                                 const uint8_t* __restrict__ wt /*[ntile][KC16][128][16] e4m3*/,
                                                                 const float* __restrict__ scale, const float* __restrict__ bias,
                                                                 int relu, uint8_t* __restrict__ out, int64_t ldo) {
    using G = L8Geo<D, DEPTH>;
    extern __shared__ __attribute__((aligned(16))) uint8_t l8[];
    uint8_t* lw = l8;
    float* lsc = reinterpret_cast<float*>(l8 + G::W_BYTES);
    float* lbi = lsc + kL8Cols;
    uint2* lut = reinterpret_cast<uint2*>(lbi + kL8Cols);  // mask byte -> its eight e4m3 bytes (1.0 = 0x38 where the bit is set)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    uint8_t* lx = l8 + G::W_BYTES + G::MISC_BYTES + wv * (4096 + G::ROW_BYTES);  // the wave's exchange slice ...
    uint8_t* lr = lx + 4096;                                                      // ... and its staged state rows
    const int64_t n0 = (int64_t)blockIdx.x * kL8Cols;
    // The wave's 64 rows of a chunk (64 x D contiguous bytes; rw is a multiple of 64, so rw * D is a multiple of 16) are
    // PREFETCHED into registers one chunk ahead: with two waves per SIMD nothing else hides the HBM round trip of the rows,
    // which is as long as the chunk's whole K loop (the first cut waited for it at the top of every chunk: 31 % MFMA-busy).
    uint4 pre[G::NPRE];
    auto prefetch = [&](int64_t chunk_n) {
        const int64_t rwn = chunk_n * kL8Threads + wv * 64;
#pragma unroll
        for (int j = 0; j < G::NPRE; j++) pre[j] = make_uint4(0u, 0u, 0u, 0u);
        if (rwn >= m) return;
        const int nb = (int)((m - rwn) < 64 ? (m - rwn) : 64) * D;
        const uint8_t* g = nn + rwn * D;
#pragma unroll
        for (int j = 0; j < G::NPRE; j++) {
            const int q = lane + 64 * j;
            if (q * 16 + 16 <= nb) {
                pre[j] = *reinterpret_cast<const uint4*>(g + q * 16);
            } else if (q * 16 < nb) {  // the last, partial piece of the matrix: byte by byte (nothing is read past its end)
                uint32_t t[4] = {0u, 0u, 0u, 0u};
                for (int b = q * 16; b < nb; b++) t[(b & 15) >> 2] |= (uint32_t)g[b] << (8 * (b & 3));
                pre[j] = make_uint4(t[0], t[1], t[2], t[3]);
            }
        }
    };
    prefetch(blockIdx.y);
    {
        const uint4* src = reinterpret_cast<const uint4*>(wt) + (size_t)blockIdx.x * (G::W_BYTES / 16);
        uint4* dst = reinterpret_cast<uint4*>(lw);
        for (int q = threadIdx.x; q < G::W_BYTES / 16; q += kL8Threads) dst[q] = src[q];
        if (threadIdx.x < kL8Cols) {
            lsc[threadIdx.x] = scale[n0 + threadIdx.x];
            lbi[threadIdx.x] = bias[n0 + threadIdx.x];
        }
        if (threadIdx.x < 256) {  // bit i of a nibble -> byte i: * 0x00204081 & 0x01010101; 1.0 in e4m3 = 0x38
            const uint32_t b = threadIdx.x;
            lut[b] = make_uint2((((b & 0xFu) * 0x00204081u) & 0x01010101u) * 0x38u, (((b >> 4) * 0x00204081u) & 0x01010101u) * 0x38u);
        }
    }
    __syncthreads();
    const int wcol = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
    for (int64_t chunk = blockIdx.y; chunk * kL8Threads < m; chunk += gridDim.y) {
        const int64_t rw = chunk * kL8Threads + wv * 64;  // first row of this wave (no workgroup barrier below: waves run free)
        if (rw >= m) continue;
        const int nrows = (int)((m - rw) < 64 ? (m - rw) : 64);
        // ---- the wave's rows (prefetched) -> LDS
        {
#pragma unroll
            for (int j = 0; j < G::NPRE; j++) {
                const int q = lane + 64 * j;
                if (q * 16 < G::ROW_BYTES - 16) *reinterpret_cast<uint4*>(lr + q * 16) = pre[j];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        // ---- one-hot mask of ONE row per lane (row l31 for h = 0, row 32 + l31 for h = 1): bit pos * DEPTH + value
        uint32_t full[G::MW];
#pragma unroll
        for (int w = 0; w < G::MW; w++) full[w] = 0;
        {
            const int myrow = h * 32 + l31;
            const uint8_t* row = lr + myrow * D;
            const bool live = myrow < nrows;
            auto put = [&](int pos, uint32_t c) {
                const int bit0 = pos * DEPTH, w0 = bit0 >> 5, sh = bit0 & 31;
                if (sh + DEPTH <= 32) {
                    full[w0] |= (1u << sh) << c;
                } else {
                    const uint64_t f = (uint64_t)1 << (c + (uint32_t)sh);
                    full[w0] |= (uint32_t)f;
                    if (w0 + 1 < G::MW) full[w0 + 1] |= (uint32_t)(f >> 32);
                }
            };
            if (live) {
                if constexpr (D % 2 == 0) {  // rows start on even offsets: two positions per LDS read
#pragma unroll
                    for (int pos = 0; pos < D; pos += 2) {
                        const uint32_t v = *reinterpret_cast<const uint16_t*>(row + pos);
                        put(pos, v & 0xFFu);
                        put(pos + 1, v >> 8);
                    }
                } else {
#pragma unroll
                    for (int pos = 0; pos < D; pos++) put(pos, row[pos]);
                }
            }
        }
        // lane h feeds the instruction of K step s with bits [64 s + 32 h, + 32) of BOTH its rows (block 0: row l31, block 1:
        // row 32 + l31) = word 2 s + h of each mask.  v_permlane32_swap(X, Y) leaves lanes 0-31 with (X, X of lane + 32) and
        // lanes 32-63 with (Y of lane - 32, Y): with X / Y the even / odd word of the lane's own row that is (mask of row l31,
        // mask of row 32 + l31), word 2 s + h, in every lane.
        uint32_t mk[2][G::KSTEPS];
#pragma unroll
        for (int s = 0; s < G::KSTEPS; s++) {
            const auto r = __builtin_amdgcn_permlane32_swap(full[2 * s], full[2 * s + 1], false, false);
            mk[0][s] = r[0];
            mk[1][s] = r[1];
        }
        __builtin_amdgcn_wave_barrier();  // (every lane has read its row: the slice may be rewritten by the next chunk)
        prefetch(chunk + gridDim.y);      // the next chunk's rows fly under this chunk's K loop and tail
        l8_f32x16 acc[2][4];
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int jn = 0; jn < 4; jn++)
#pragma unroll
                for (int e = 0; e < 16; e++) acc[i][jn][e] = 0.f;
#pragma unroll
        for (int s = 0; s < G::KSTEPS; s++) {
            l8_i32x8 bf[2];
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int d = 0; d < 4; d++) {  // mask byte d -> eight e4m3 bytes: one 8-byte LDS read (the arithmetic form — two
                    const uint2 t = lut[(mk[i][s] >> (8 * d)) & 0xFFu];  // multiplies and two masks per four bytes — was a third of the kernel's VALU work)
                    bf[i][2 * d] = (int)t.x;
                    bf[i][2 * d + 1] = (int)t.y;
                }
#pragma unroll
            for (int jn = 0; jn < 4; jn++) {
                const uint8_t* p = lw + ((size_t)((s * 4 + h * 2) * kL8Cols + jn * 32 + wcol)) * 16;
                const l8_i32x4 w0 = *reinterpret_cast<const l8_i32x4*>(p);
                const l8_i32x4 w1 = *reinterpret_cast<const l8_i32x4*>(p + kL8Cols * 16);
                const l8_i32x8 wf = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
                acc[0][jn] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wf, bf[0], acc[0][jn], 0, 0, 0, 0, 0, 0);
                acc[1][jn] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wf, bf[1], acc[1][jn], 0, 0, 0, 0, 0, 0);
            }
        }
        // ---- tail.  lane (l31, h), block (i, jn), registers 8 x .. 8 x + 7  ->  row rw + 32 i + l31, columns
        // n0 + jn * 32 + x * 16 + h * 8 .. + 8: the 8-byte piece c8 = jn * 4 + x * 2 + h of the row's 128 bytes
        const int tg = l31 >> 2, tx = l31 & 3;
#pragma unroll
        for (int i = 0; i < 2; i++) {
#pragma unroll
            for (int jn = 0; jn < 4; jn++)
#pragma unroll
                for (int x = 0; x < 2; x++) {
                    const int cb = jn * 32 + x * 16 + h * 8;
                    const float4 s0 = *reinterpret_cast<const float4*>(lsc + cb), s1 = *reinterpret_cast<const float4*>(lsc + cb + 4);
                    const float4 b0 = *reinterpret_cast<const float4*>(lbi + cb), b1 = *reinterpret_cast<const float4*>(lbi + cb + 4);
                    const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
                    const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
                    float u[8];
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        const float v = fmaf(acc[i][jn][8 * x + e], sc[e], bb[e]);
                        // e4m3fn has no infinity: saturate at +-448 (with ReLU the lower bound is 0: one v_med3_f32)
                        u[e] = __builtin_amdgcn_fmed3f(v, relu ? 0.f : -448.f, 448.f);
                    }
                    uint32_t q0 = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(u[0], u[1], 0, false);
                    q0 = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(u[2], u[3], (int)q0, true);
                    uint32_t q1 = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(u[4], u[5], 0, false);
                    q1 = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(u[6], u[7], (int)q1, true);
                    const int c8 = jn * 4 + x * 2 + h;
                    *reinterpret_cast<uint2*>(lx + l31 * 128 + (((c8 >> 1) ^ (l31 & 7)) << 4) + ((c8 & 1) << 3)) = make_uint2(q0, q1);
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private slice: only the wave's own writes
            __builtin_amdgcn_wave_barrier();
            // lane (4 g + t, h) leaves with the 16-byte piece t * 2 + h of rows 4 g .. 4 g + 3: 8 lanes per 128-byte row
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int row = 4 * tg + j, c16 = tx * 2 + h;
                const uint4 q = *reinterpret_cast<const uint4*>(lx + row * 128 + ((c16 ^ (row & 7)) << 4));
                const int64_t r = rw + 32 * i + row;
                if (r < m) *reinterpret_cast<uint4*>(out + r * ldo + n0 + c16 * 16) = q;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the reads are done before the slice is rewritten
            __builtin_amdgcn_wave_barrier();
        }
    }
}

template <int D, int DEPTH>
int launch_l1_8(const uint8_t* nn, int64_t m, const uint8_t* wt, const float* scale, const float* bias, int relu, uint8_t* out,
                int64_t n_pad, hipStream_t s) {
    using G = L8Geo<D, DEPTH>;
    static_assert(G::LDS <= 160 * 1024, "weight tile does not fit LDS");
    const int64_t chunks = (m + kL8Threads - 1) / kL8Threads;
    const int64_t tiles = n_pad / kL8Cols;
    // row chunks are dealt to gridDim.y workgroups per column tile; 1280 workgroups (5 per CU) where the tile count divides it
    int64_t gy = 1280 / tiles;
    if (gy < 1) gy = 1;
    if (gy > chunks) gy = chunks;
    auto kern = k_l1_onehot_gemm8<D, DEPTH>;
    DCA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles, (unsigned)gy), dim3(kL8Threads), G::LDS, s, nn, m, wt, scale, bias, relu, out, n_pad);
    return launch_check("k_l1_onehot_gemm8");
}

}  // namespace dca

using namespace dca;

extern "C" {

// geometries whose whole weight tile (128 columns x KPAD bytes) fits LDS next to the waves' slices: cube3 (384: 48 KB),
// puzzle15 (256: 32 KB), puzzle24 (640: 80 KB), lightsout7 (320: 40 KB)
int dca_l1_supported8(int state_dim, int depth) {
    return (state_dim == 54 && depth == 6) || (state_dim == 16 && depth == 16) || (state_dim == 25 && depth == 25) ||
           (state_dim == 49 && depth == 6);
}

int64_t dca_l1_kpad8(int state_dim, int depth) { return (((int64_t)state_dim * depth + 63) / 64) * 64; }

int dca_l1_onehot_gemm8(const uint8_t* nnet_in, int64_t m, int state_dim, int depth, const void* w_tiles, int64_t n_pad,
                        const float* scale, const float* bias, int relu, void* out8, void* stream) {
    DCA_ARG(nnet_in && w_tiles && scale && bias && out8 && m >= 0 && n_pad >= kL8Cols && n_pad % kL8Cols == 0);
    DCA_ARG((reinterpret_cast<uintptr_t>(nnet_in) & 15) == 0 && (reinterpret_cast<uintptr_t>(out8) & 15) == 0);
    if (!dca_l1_supported8(state_dim, depth)) {
        set_error("dca_l1_onehot_gemm8: geometry (%d, %d) not instantiated (weight tile must fit LDS)", state_dim, depth);
        return DCA_E_BADARG;
    }
    if (m == 0) return 0;
    const uint8_t* wt = reinterpret_cast<const uint8_t*>(w_tiles);
    uint8_t* o = reinterpret_cast<uint8_t*>(out8);
    hipStream_t s = (hipStream_t)stream;
    if (state_dim == 54) return launch_l1_8<54, 6>(nnet_in, m, wt, scale, bias, relu, o, n_pad, s);
    if (state_dim == 16) return launch_l1_8<16, 16>(nnet_in, m, wt, scale, bias, relu, o, n_pad, s);
    if (state_dim == 25) return launch_l1_8<25, 25>(nnet_in, m, wt, scale, bias, relu, o, n_pad, s);
    return launch_l1_8<49, 6>(nnet_in, m, wt, scale, bias, relu, o, n_pad, s);
}

}  // extern "C"
