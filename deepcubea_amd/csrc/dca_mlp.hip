// dca_mlp.hip — first layer of the cost-to-go network as a hand-written MFMA kernel for gfx950.
//
// Reference arithmetic (utils/pytorch_models.py:49-60): x = one_hot(states_nnet, depth).float().view(-1, D*depth);
// y = relu(bn1(fc1(x))).  With BatchNorm folded (eval statistics) this is  y = relu(onehot(s) . W1^T + b1),  a GEMM whose
// A operand has exactly D ones per row.  What the kernel exploits, and a library GEMM cannot:
//
//   * the one-hot matrix is never materialised: a lane rebuilds its A fragments from the D colour / tile bytes of its
//     row (a K-bit mask in registers, 8 bits -> 8 bf16 per MFMA operand);
//   * A is exactly representable in bf16, so splitting the fp32 weights into P bf16 planes (W = hi + mid + lo, 8+8+8
//     mantissa bits) makes every MFMA product exact and the fp32-accumulated result an fp32 GEMM — on the bf16 MFMA
//     pipes, which are 16x faster than the f32-input MFMA the library's fp32 GEMM has to use.  P = 3 is the fp32
//     parity mode, P = 2 serves fp16 weights, P = 1 bf16;
//   * bias + ReLU + the output cast ride in the epilogue.
//
// Tiling: a workgroup (8 waves) owns 64 output columns.  Their weights — all P planes — are staged into LDS (cube3:
// all of K once, 129 KB with P = 3; the sliding puzzles, K up to 2401: 320 one-hot columns at a time, walked per row
// chunk) in the exact order the B fragments are read
// ([plane][k/8][column][8] => 512 contiguous bytes per half-wave, conflict-free), then the workgroup walks over row
// chunks of 512 (64 rows per wave = 2x2 tiles of v_mfma_f32_32x32x16_bf16, 64 accumulator VGPRs).  Per K-step a wave
// issues 4*P MFMAs for 2 A-fragment rebuilds and 2*P ds_read_b128: the MFMA pipe is the limiter.
#include "dca_common.h"

namespace dca {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int kL1Threads = 512;  // 8 waves = 2 per SIMD: one wave's mask build / epilogue hides under the other's MFMAs
constexpr int kL1Rows = kL1Threads;  // rows per chunk (64 per wave)

template <int D, int DEPTH>
struct L1Geo {
    static constexpr int K = D * DEPTH;
    static constexpr int KSTEPS = (K + 15) / 16;  // MFMA K = 16
    static constexpr int KPAD = KSTEPS * 16;
    static constexpr int KC = KPAD / 8;           // 16-byte weight chunks along K
    // K is walked in chunks whose weights (all planes) fit LDS: the whole of K where it fits in one piece (cube3: 21
    // steps = 129 KB with 3 planes, staged once per workgroup), else 20 steps = 320 one-hot columns (120 KB) at a time,
    // re-staged for every 512-row chunk (928 KB of L2 reads against 472 MFLOP of MFMA work for puzzle48: noise)
    static constexpr int CH_STEPS = KSTEPS <= 21 ? KSTEPS : 20;
    static constexpr int NCH = (KSTEPS + CH_STEPS - 1) / CH_STEPS;
    static constexpr int CHKC = CH_STEPS * 2;     // 16-byte weight chunks per LDS chunk and plane
    static constexpr int MWC = (CH_STEPS * 16 + 31) / 32;  // one-hot mask words per row and chunk
};

__device__ __forceinline__ uint16_t f32_to_bf16_rne(float f) {  // gfx950's v_cvt_pk_bf16_f32 (round to nearest even)
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 v = {f, 0.f};
    const bf2 b = __builtin_convertvector(v, bf2);
    uint32_t u;
    __builtin_memcpy(&u, &b, 4);
    return (uint16_t)(u & 0xFFFFu);
}

template <int D, int DEPTH, int P,
          int OUT /*0 f32, 1 f16, 2 bf16, 3 f16x3 split (vh, vl, vh) interleaved, 4 two fp16 planes (high, then low at +m*ldo),
                    5 e4m3 bytes (saturating), 6 e4m3 bytes with one E8M0 block scale per row and 64 columns (out_scale)*/>
__global__ __launch_bounds__(kL1Threads) void k_l1_onehot_gemm(const uint8_t* __restrict__ nn, int64_t m,
                                                        const uint8_t* __restrict__ wt /*[ntile][P][KC][64][8] bf16*/,
                                                        const float* __restrict__ bias, int relu, void* __restrict__ out,
                                                        int64_t ldo, int* __restrict__ overflow,
                                                        uint8_t* __restrict__ out_scale /*OUT 6: [m, ld_sc]*/, int64_t ld_sc) {
    using G = L1Geo<D, DEPTH>;
    extern __shared__ __attribute__((aligned(16))) uint8_t lw[];
    // weights of K-chunk ch (all planes) -> LDS, in B-fragment order [plane][k/8][column][8]
    auto stage = [&](int ch) {
        const int kc0 = ch * G::CHKC;
        const int nkc = (G::KC - kc0) < G::CHKC ? (G::KC - kc0) : G::CHKC;
#pragma unroll
        for (int p = 0; p < P; p++) {
            const uint4* src = reinterpret_cast<const uint4*>(wt) + ((size_t)(blockIdx.x * P + p) * G::KC + kc0) * 64;
            uint4* dst = reinterpret_cast<uint4*>(lw) + (size_t)p * G::CHKC * 64;
            for (int q = threadIdx.x; q < nkc * 64; q += kL1Threads) dst[q] = src[q];
        }
    };
    if constexpr (G::NCH == 1) {
        stage(0);
        __syncthreads();
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int64_t n0 = (int64_t)blockIdx.x * 64;
    // LEAN (round 5; 16-bit and e4m3 outputs): the MFMA's operand roles are swapped — weights = A operand, one-hot rows = B
    // operand — so that a lane holds a piece of ONE output row, and the weight column that feeds MFMA row i is chosen (bits 2
    // and 3 of the fragment's column index swapped: the same 32 chunks per lane group, still conflict-free) such that a
    // lane's 8 consecutive accumulator registers are 8 CONSECUTIVE output columns: bias, ReLU and the rounding work on
    // register octets, packed pieces change hands through the wave's 4 KB of LDS, 8 lanes store one row segment.  The
    // accumulator-layout tail it replaces (one LDS word per element: 128 ds_write_b32 + 16 ds_read_b128 + 16 eight-byte
    // stores per wave and 64 x 64 tile) was half the kernel's time with one weight plane (csrc/dca_gemm16.hip has the story).
    constexpr bool LEAN = OUT == 1 || OUT == 2 || OUT == 4 || OUT == 5;
    const int wcol = LEAN ? ((l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1)) : l31;
    float bv[2];
    bv[0] = bias[n0 + l31];
    bv[1] = bias[n0 + 32 + l31];
    // LEAN: the bias of a lane's 4 x 8 columns (piece X: columns X * 16 + h * 8 .. + 8).  With ONE weight plane the kernel must
    // stay at 128 VGPRs — two workgroups per CU: 0.92 vs 1.08 ms per 204 800 x 5120 layer — so the bias is fetched per piece in
    // the tail (two 16-byte loads that hit L1, hidden by the other workgroup); with two or three planes the weight tile leaves room
    // for one workgroup per CU anyway and the 32 registers are free: held across the K loop (per-piece loads: 2.47 vs 2.09 ms).
    constexpr bool BIAS_REGS = LEAN && P > 1;
    float bq[BIAS_REGS ? 4 : 1][8];
    if constexpr (BIAS_REGS) {
#pragma unroll
        for (int X = 0; X < 4; X++)
#pragma unroll
            for (int e = 0; e < 8; e++) bq[X][e] = bias[n0 + X * 16 + h * 8 + e];
    }
    for (int64_t chunk = blockIdx.y; chunk * kL1Rows < m; chunk += gridDim.y) {
        const int64_t rw = chunk * kL1Rows + wv * 64;  // first row of this wave
        // (with K-chunking every wave takes part in the staging barriers, rows or not)
        if (G::NCH == 1 && rw >= m) continue;
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int jn = 0; jn < 2; jn++)
#pragma unroll
                for (int e = 0; e < 16; e++) acc[i][jn][e] = 0.f;
#pragma unroll 1
        for (int ch = 0; ch < G::NCH; ch++) {
            if constexpr (G::NCH > 1) {
                __syncthreads();  // the previous chunk's readers are done with the LDS tile
                stage(ch);
                __syncthreads();
            }
            const int steps = (G::KSTEPS - ch * G::CH_STEPS) < G::CH_STEPS ? (G::KSTEPS - ch * G::CH_STEPS) : G::CH_STEPS;
            // one-hot mask of this lane's two rows, restricted to the chunk's columns [k0, k0 + 16*steps)
            uint32_t mk[2][G::MWC];
#pragma unroll
            for (int i = 0; i < 2; i++) {
#pragma unroll
                for (int w = 0; w < G::MWC; w++) mk[i][w] = 0;
                const int64_t r = rw + 32 * i + l31;
                if (r < m) {
                    const uint8_t* row = nn + r * D;
                    if constexpr (G::NCH == 1) {
#pragma unroll
                        for (int pos = 0; pos < D; pos++) {
                            const uint32_t c = row[pos];
                            const int bit0 = pos * DEPTH, w0 = bit0 >> 5, sh = bit0 & 31;
                            const uint64_t f = (uint64_t)1 << (c + (uint32_t)sh);
                            mk[i][w0] |= (uint32_t)f;
                            if (w0 + 1 < G::MWC) mk[i][w0 + 1] |= (uint32_t)(f >> 32);
                        }
                    } else {
                        // only the positions whose DEPTH columns overlap the chunk (~ 320 / DEPTH + 2 of them)
                        const int k0 = ch * G::CH_STEPS * 16, k1 = k0 + steps * 16;
                        const int p_lo = k0 / DEPTH, p_hi = (k1 - 1) / DEPTH < D - 1 ? (k1 - 1) / DEPTH : D - 1;
                        for (int pos = p_lo; pos <= p_hi; pos++) {
                            const int bit = pos * DEPTH + (int)row[pos] - k0;
                            const uint32_t wsel = (uint32_t)(bit >> 5), bm = 1u << (bit & 31);
                            const bool in = bit >= 0 && bit < steps * 16;
#pragma unroll
                            for (int w = 0; w < G::MWC; w++) mk[i][w] |= (in && wsel == (uint32_t)w) ? bm : 0u;
                        }
                    }
                }
            }
#pragma unroll
            for (int s = 0; s < G::CH_STEPS; s++) {
                if (G::NCH > 1 && s >= steps) break;
                bf16x8 a[2];
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    const uint32_t byte = (mk[i][s >> 1] >> ((s & 1) * 16 + 8 * h)) & 0xFFu;
                    uint32_t v[4];
#pragma unroll
                    for (int j = 0; j < 4; j++)  // bits (2j, 2j+1) -> two bf16 ones: spread to bits 0 and 16, scale by 0x3F80
                        v[j] = ((((byte >> (2 * j)) & 3u) * 0x8001u) & 0x00010001u) * 0x3F80u;
                    __builtin_memcpy(&a[i], v, 16);
                }
#pragma unroll
                for (int p = 0; p < P; p++) {
#pragma unroll
                    for (int jn = 0; jn < 2; jn++) {
                        const bf16x8 b = *reinterpret_cast<const bf16x8*>(
                            lw + ((size_t)((p * G::CHKC + 2 * s + h) * 64 + jn * 32 + wcol)) * 16);
                        if constexpr (LEAN) {  // weights = the instruction's A operand: D[i][j] has j = lane & 31 = the state's row
                            acc[0][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a[0], acc[0][jn], 0, 0, 0);
                            acc[1][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a[1], acc[1][jn], 0, 0, 0);
                        } else {
                            acc[0][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b, acc[0][jn], 0, 0, 0);
                            acc[1][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b, acc[1][jn], 0, 0, 0);
                        }
                    }
                }
            }
        }
        if constexpr (LEAN) {
            // lane (l31, h), block (i, jn), registers 8 x .. 8 x + 7  ->  row rw + 32 i + l31, columns n0 + (jn * 2 + x) * 16 + h * 8 .. + 8
            uint8_t* tl = lw + G::CHKC * P * 64 * 16 + wv * 4096;
            const int tg = l31 >> 2, tx = l31 & 3;
            constexpr int PB = OUT == 5 ? 8 : 16;        // bytes of a packed piece (8 values)
            constexpr int RB = 8 * PB;                   // bytes of a row of the wave's 64 columns
            auto addr = [&](int row, int c16) { return tl + row * RB + ((c16 ^ (row & 7)) * PB); };
            bool ovf = false;
            constexpr int NPASS = OUT == 4 ? 2 : 1;      // two fp16 planes: the high halves, then the low halves
#pragma unroll
            for (int i = 0; i < 2; i++) {
#pragma unroll
                for (int pass = 0; pass < NPASS; pass++) {
#pragma unroll
                    for (int X = 0; X < 4; X++) {
                        float bb[8];
                        if constexpr (BIAS_REGS) {
#pragma unroll
                            for (int e = 0; e < 8; e++) bb[e] = bq[X][e];
                        } else {
                            const float4 b0 = *reinterpret_cast<const float4*>(bias + n0 + X * 16 + h * 8);
                            const float4 b1 = *reinterpret_cast<const float4*>(bias + n0 + X * 16 + h * 8 + 4);
                            bb[0] = b0.x, bb[1] = b0.y, bb[2] = b0.z, bb[3] = b0.w, bb[4] = b1.x, bb[5] = b1.y, bb[6] = b1.z, bb[7] = b1.w;
                        }
                        float u[8];
#pragma unroll
                        for (int e = 0; e < 8; e++) {
                            float v = acc[i][X >> 1][8 * (X & 1) + e] + bb[e];
                            if (relu) v = fmaxf(v, 0.f);
                            if constexpr (OUT == 4) ovf |= !(fabsf(v) <= 60000.0f);
                            u[e] = v;
                        }
                        if constexpr (OUT == 5) {  // e4m3fn has no infinity: saturate at +-448
                            uint32_t w0 = 0, w1 = 0;
                            auto sat = [](float f) { return fminf(fmaxf(f, -448.f), 448.f); };
                            w0 = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(sat(u[0]), sat(u[1]), 0, false);
                            w0 = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(sat(u[2]), sat(u[3]), (int)w0, true);
                            w1 = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(sat(u[4]), sat(u[5]), 0, false);
                            w1 = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(sat(u[6]), sat(u[7]), (int)w1, true);
                            *reinterpret_cast<uint2*>(addr(l31, X * 2 + h)) = make_uint2(w0, w1);
                        } else {
                            uint32_t w[4];
#pragma unroll
                            for (int e = 0; e < 4; e++) {
                                const float a0 = u[2 * e], a1 = u[2 * e + 1];
                                if constexpr (OUT == 2) {
                                    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
                                    typedef float f2 __attribute__((ext_vector_type(2)));
                                    const f2 vv = {a0, a1};
                                    const bf2 bb2 = __builtin_convertvector(vv, bf2);
                                    __builtin_memcpy(&w[e], &bb2, 4);
                                } else {
                                    _Float16 h0 = (_Float16)a0, h1 = (_Float16)a1;
                                    if (OUT == 4 && pass == 1) {  // the low plane: what the high halves left over
                                        h0 = (_Float16)(a0 - (float)h0);
                                        h1 = (_Float16)(a1 - (float)h1);
                                    }
                                    uint16_t c0, c1;
                                    __builtin_memcpy(&c0, &h0, 2);
                                    __builtin_memcpy(&c1, &h1, 2);
                                    w[e] = (uint32_t)c0 | ((uint32_t)c1 << 16);
                                }
                            }
                            *reinterpret_cast<uint4*>(addr(l31, X * 2 + h)) = make_uint4(w[0], w[1], w[2], w[3]);
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private slice: only the wave's own writes
                    // lane (4 g + x, h) leaves with piece x * 2 + h of rows 4 g .. 4 g + 3: 8 lanes per row segment
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int64_t r = rw + 32 * i + 4 * tg + j;
                        const int c16 = tx * 2 + h;
                        if constexpr (OUT == 5) {
                            const uint2 q = *reinterpret_cast<const uint2*>(addr(4 * tg + j, c16));
                            if (r < m) *reinterpret_cast<uint2*>(reinterpret_cast<uint8_t*>(out) + r * ldo + n0 + c16 * 8) = q;
                        } else {
                            const uint4 q = *reinterpret_cast<const uint4*>(addr(4 * tg + j, c16));
                            uint16_t* q0 = reinterpret_cast<uint16_t*>(out) + r * ldo + n0 + c16 * 8 + (pass ? m * ldo : 0);
                            if (r < m) *reinterpret_cast<uint4*>(q0) = q;
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the reads are done before the slice is rewritten
                }
            }
            if (OUT == 4 && ovf && overflow) *overflow = 1;
        } else
        // epilogue: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
        if constexpr (OUT == 1 || OUT == 2 || OUT == 4 || OUT == 5 || OUT == 6) {
            // 16-bit outputs: straight from the accumulator layout every store instruction would write 2 bytes per lane,
            // 64 contiguous bytes per row (3.1 ms for the 204 800 x 5120 planes, the fp32 output of the same tile 2.3 ms).
            // Each wave transposes 16 rows x 64 columns at a time through 4 KB of its own LDS — one word per element:
            // value in the low half, split residual (planes) in the high half — and leaves with 8-byte stores: 16 lanes
            // cover a row's 128 contiguous bytes of a plane.
            uint32_t* sl = reinterpret_cast<uint32_t*>(lw + G::CHKC * P * 64 * 16 + wv * 4096);
            typedef uint16_t u16x4 __attribute__((ext_vector_type(4)));
            bool ovf = false;
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int half = 0; half < 2; half++) {
#pragma unroll
                    for (int jn = 0; jn < 2; jn++)
#pragma unroll
                        for (int rr = 0; rr < 8; rr++) {
                            const int reg = 8 * half + rr;
                            float v = acc[i][jn][reg] + bv[jn];
                            if (relu) v = fmaxf(v, 0.f);
                            uint32_t w;
                            if constexpr (OUT == 2) {
                                w = f32_to_bf16_rne(v);
                            } else if constexpr (OUT == 5) {  // e4m3fn has no infinity: saturate at +-448
                                w = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(v, -448.f), 448.f), 0.f, 0, false) & 0xFFu;
                            } else if constexpr (OUT == 6) {  // quantised after the transposition, once the row's block scale is known
                                w = __float_as_uint(v);
                            } else {
                                const _Float16 hh = (_Float16)v;
                                uint16_t hb, lb = 0;
                                __builtin_memcpy(&hb, &hh, 2);
                                if constexpr (OUT == 4) {
                                    ovf |= !(fabsf(v) <= 60000.0f);
                                    const _Float16 ll = (_Float16)(v - (float)hh);
                                    __builtin_memcpy(&lb, &ll, 2);
                                }
                                w = (uint32_t)hb | ((uint32_t)lb << 16);
                            }
                            sl[((rr & 3) + 8 * (rr >> 2) + 4 * h) * 64 + jn * 32 + l31] = w;
                        }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private slice: only the wave's own writes
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int rl = q * 4 + (lane >> 4), c4 = (lane & 15) * 4;
                        const uint4 pk = *reinterpret_cast<const uint4*>(sl + rl * 64 + c4);
                        const int64_t r = rw + 32 * i + 16 * half + rl;
                        if constexpr (OUT == 5) {  // one byte per element: 4-byte stores, 16 lanes cover 64 contiguous bytes of a row
                            if (r < m)
                                *reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(out) + r * ldo + n0 + c4) =
                                    (pk.x & 0xFFu) | ((pk.y & 0xFFu) << 8) | ((pk.z & 0xFFu) << 16) | (pk.w << 24);
                        } else if constexpr (OUT == 6) {
                            // the workgroup's 64 columns ARE one scale block of the next layer's K: largest magnitude of the row's 64
                            // values (16 lanes x 4), the power of two that brings it to e4m3's range, then the bytes
                            const float u0 = __uint_as_float(pk.x), u1 = __uint_as_float(pk.y), u2 = __uint_as_float(pk.z),
                                        u3 = __uint_as_float(pk.w);
                            float am = fmaxf(fmaxf(fabsf(u0), fabsf(u1)), fmaxf(fabsf(u2), fabsf(u3)));
                            // (over the DPP row of 16 lanes that holds this row's 64 columns: rotate-and-max on the VALU)
                            am = fmaxf(am, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(am), 0x128, 0xF, 0xF, false)));
                            am = fmaxf(am, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(am), 0x124, 0xF, 0xF, false)));
                            am = fmaxf(am, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(am), 0x122, 0xF, 0xF, false)));
                            am = fmaxf(am, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(am), 0x121, 0xF, 0xF, false)));
                            const float t = am * (1.0f / 448.0f);
                            const uint32_t tb = __float_as_uint(t);
                            int e = (int)((tb >> 23) & 0xFFu) - 127 + ((tb & 0x7FFFFFu) ? 1 : 0);
                            e = e < -126 ? -126 : (e > 126 ? 126 : e);
                            const float inv = __uint_as_float((uint32_t)(127 - e) << 23);
                            if (r < m) {
                                uint32_t q8 = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(fminf(u0 * inv, 448.f), fminf(u1 * inv, 448.f), 0, false);
                                q8 = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(fminf(u2 * inv, 448.f), fminf(u3 * inv, 448.f), (int)q8, true);
                                *reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(out) + r * ldo + n0 + c4) = q8;
                                if ((lane & 15) == 0) out_scale[r * ld_sc + blockIdx.x] = (uint8_t)(e + 127);
                            }
                        } else if (r < m) {
                            const u16x4 hi = {(uint16_t)pk.x, (uint16_t)pk.y, (uint16_t)pk.z, (uint16_t)pk.w};
                            uint16_t* q0 = reinterpret_cast<uint16_t*>(out) + r * ldo + n0 + c4;
                            *reinterpret_cast<u16x4*>(q0) = hi;
                            if constexpr (OUT == 4) {
                                const u16x4 lo = {(uint16_t)(pk.x >> 16), (uint16_t)(pk.y >> 16), (uint16_t)(pk.z >> 16),
                                                  (uint16_t)(pk.w >> 16)};
                                *reinterpret_cast<u16x4*>(q0 + m * ldo) = lo;
                            }
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the reads are done before the slice is rewritten
                }
            if (OUT == 4 && ovf && overflow) *overflow = 1;
        } else
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int jn = 0; jn < 2; jn++) {
                const int64_t col = n0 + jn * 32 + l31;
#pragma unroll
                for (int reg = 0; reg < 16; reg++) {
                    const int64_t r = rw + 32 * i + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                    if (r < m) {
                        float v = acc[i][jn][reg] + bv[jn];
                        if (relu) v = fmaxf(v, 0.f);
                        if constexpr (OUT == 0)
                            reinterpret_cast<float*>(out)[r * ldo + col] = v;
                        else if constexpr (OUT == 1)
                            reinterpret_cast<_Float16*>(out)[r * ldo + col] = (_Float16)v;
                        else if constexpr (OUT == 2)
                            reinterpret_cast<uint16_t*>(out)[r * ldo + col] = f32_to_bf16_rne(v);
                        else if constexpr (OUT == 3) {  // the library-GEMM f16x3 A operand: 192 contiguous bytes per 32 lanes
                            _Float16* q = reinterpret_cast<_Float16*>(out) + (r * ldo + col) * 3;
                            if (!(fabsf(v) <= 60000.0f) && overflow) *overflow = 1;
                            const _Float16 hh = (_Float16)v;
                            q[0] = hh;
                            q[1] = (_Float16)(v - (float)hh);
                            q[2] = hh;
                        } else {  // the two fp16 planes dca_f16x3_gemm reads: high halves, then (m*ldo further) low halves
                            _Float16* q = reinterpret_cast<_Float16*>(out) + r * ldo + col;
                            if (!(fabsf(v) <= 60000.0f) && overflow) *overflow = 1;
                            const _Float16 hh = (_Float16)v;
                            q[0] = hh;
                            q[m * ldo] = (_Float16)(v - (float)hh);
                        }
                    }
                }
            }
    }
}

template <int D, int DEPTH, int P>
int launch_l1_out(const uint8_t* nn, int64_t m, const uint8_t* wt, const float* bias, int relu, void* out, int out_dtype,
                  int64_t n_pad, int* overflow, hipStream_t s, uint8_t* out_scale = nullptr, int64_t ld_sc = 0) {
    using G = L1Geo<D, DEPTH>;
    constexpr int LDS = P * G::CHKC * 64 * 16 + (kL1Threads / 64) * 4096;  // weight tile + the waves' epilogue slices
    static_assert(LDS <= 160 * 1024, "weight tile does not fit LDS");
    const int64_t chunks = (m + kL1Rows - 1) / kL1Rows;
    const dim3 grid((unsigned)(n_pad / 64), (unsigned)(chunks < 16 ? (chunks < 1 ? 1 : chunks) : 16)), block(kL1Threads);
#define DCA_L1_LAUNCH(OUTV)                                                                                          \
    do {                                                                                                             \
        auto kern = k_l1_onehot_gemm<D, DEPTH, P, OUTV>;                                                             \
        DCA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS)); \
        hipLaunchKernelGGL(kern, grid, block, LDS, s, nn, m, wt, bias, relu, out, n_pad, overflow, out_scale, ld_sc);          \
    } while (0)
    if (out_dtype == DCA_DT_F32)
        DCA_L1_LAUNCH(0);
    else if (out_dtype == DCA_DT_F16)
        DCA_L1_LAUNCH(1);
    else if (out_dtype == DCA_DT_BF16)
        DCA_L1_LAUNCH(2);
    else if (out_dtype == DCA_DT_F16X3)
        DCA_L1_LAUNCH(3);
    else if (out_dtype == DCA_DT_E4M3) {
        if constexpr (P == 1) {  // (the fp8 mode keeps layer 1's weights as ONE bf16 plane: no other combination is built)
            if (out_scale != nullptr)
                DCA_L1_LAUNCH(6);
            else
                DCA_L1_LAUNCH(5);
        } else {
            set_error("dca_l1_onehot_gemm: e4m3 output takes planes == 1");
            return DCA_E_BADARG;
        }
    } else
        DCA_L1_LAUNCH(4);
#undef DCA_L1_LAUNCH
    return launch_check("k_l1_onehot_gemm");
}

template <int D, int DEPTH>
int launch_l1(int planes, const uint8_t* nn, int64_t m, const uint8_t* wt, const float* bias, int relu, void* out,
              int out_dtype, int64_t n_pad, int* overflow, hipStream_t s, uint8_t* out_scale = nullptr, int64_t ld_sc = 0) {
    switch (planes) {
        case 1: return launch_l1_out<D, DEPTH, 1>(nn, m, wt, bias, relu, out, out_dtype, n_pad, overflow, s, out_scale, ld_sc);
        case 2: return launch_l1_out<D, DEPTH, 2>(nn, m, wt, bias, relu, out, out_dtype, n_pad, overflow, s);
        default: return launch_l1_out<D, DEPTH, 3>(nn, m, wt, bias, relu, out, out_dtype, n_pad, overflow, s);
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// fp32-accurate dense layers on the f16 MFMA pipes ("f16x3").  An fp32 value splits exactly into two fp16 numbers to
// 22 bits (x = xh + xl, xh = f16(x), xl = f16(x - xh)); with the weights split the same way (pre-scaled by a power of two
// so their low parts stay normal)  x.w = xh.wh + xl.wh + xh.wl + O(2^-22 |x.w|).  Laid out along K as A3[3k..3k+2] =
// (xh, xl, xh) and W3[3k..3k+2] = (wh, wh, wl), ONE library f16 GEMM with fp32 output is an fp32-accurate GEMM at 3x the f16 cost — 2.4-2.9x
// faster than the library's fp32 GEMM (f32-input MFMA runs at 1/16 of the f16 rate), same error class (measured
// 1.7e-6 vs 1.2e-6 max-relative at K = 1024).  This kernel is the glue between two such GEMMs: it applies what follows
// the Linear in the network (scale back — per output unit, the weight rows carry their own power-of-two scale —, bias,
// residual add, ReLU — utils/pytorch_models.py:57-86 with BatchNorm folded)
// and emits the next layer's A3 in one pass (read 4-8 B, write 6-10 B per element).
// ---------------------------------------------------------------------------------------------------------------------
constexpr float kF16Safe = 60000.0f;  // |v| above this cannot be split into fp16 halves (fp16 max 65504)

__global__ __launch_bounds__(256) void k_act_split(const float* __restrict__ y, const float* __restrict__ bias,
                                                   const float* __restrict__ skip, const float* __restrict__ col_scale,
                                                   float alpha, int relu, int64_t m, int64_t n,
                                                   float* __restrict__ x_out /*[m,n] or null*/,
                                                   _Float16* __restrict__ a3 /*[m,3n], or two planes [2][m][n]*/,
                                                   int planes, int* __restrict__ overflow) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t col = ((int64_t)blockIdx.x * 64 + lane) * 4;
    if (col >= n) return;
    const int64_t rows_per = (m + gridDim.y - 1) / gridDim.y;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per, r1 = r0 + rows_per < m ? r0 + rows_per : m;
    float b[4] = {0.f, 0.f, 0.f, 0.f}, al[4] = {alpha, alpha, alpha, alpha};
    if (bias) {
        const float4 t = *reinterpret_cast<const float4*>(bias + col);
        b[0] = t.x, b[1] = t.y, b[2] = t.z, b[3] = t.w;
    }
    if (col_scale) {  // per-output-unit power-of-two scale of the weight rows (exact)
        const float4 t = *reinterpret_cast<const float4*>(col_scale + col);
        al[0] *= t.x, al[1] *= t.y, al[2] *= t.z, al[3] *= t.w;
    }
    typedef __attribute__((ext_vector_type(4))) _Float16 h4;
    for (int64_t r = r0 + wv; r < r1; r += 4) {
        const float4 t = *reinterpret_cast<const float4*>(y + r * n + col);
        float v[4] = {t.x, t.y, t.z, t.w};
        float sk[4] = {0.f, 0.f, 0.f, 0.f};
        if (skip) {
            const float4 q = *reinterpret_cast<const float4*>(skip + r * n + col);
            sk[0] = q.x, sk[1] = q.y, sk[2] = q.z, sk[3] = q.w;
        }
        h4 hi, lo;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            float u = v[k] * al[k] + b[k] + sk[k];
            if (relu) u = fmaxf(u, 0.f);
            v[k] = u;
            if (!(fabsf(u) <= kF16Safe) && overflow) *overflow = 1;  // beyond fp16 (or NaN): the caller redoes the batch in fp32
            const _Float16 hh = (_Float16)u;
            hi[k] = hh;
            lo[k] = (_Float16)(u - (float)hh);
        }
        if (x_out) *reinterpret_cast<float4*>(x_out + r * n + col) = make_float4(v[0], v[1], v[2], v[3]);
        if (a3 && planes) {  // dca_f16x3_gemm's operand: the high halves [m,n], then the low halves [m,n]
            *reinterpret_cast<h4*>(a3 + r * n + col) = hi;
            *reinterpret_cast<h4*>(a3 + (m + r) * n + col) = lo;
        } else if (a3) {  // element k of the row -> halves 3k..3k+2 = (vh, vl, vh): 24 contiguous bytes per lane
            h4* row = reinterpret_cast<h4*>(a3 + (r * n + col) * 3);
            h4 q0, q1, q2;
            q0[0] = hi[0], q0[1] = lo[0], q0[2] = hi[0], q0[3] = hi[1];
            q1[0] = lo[1], q1[1] = hi[1], q1[2] = hi[2], q1[3] = lo[2];
            q2[0] = hi[2], q2[1] = hi[3], q2[2] = lo[3], q2[3] = hi[3];
            row[0] = q0;
            row[1] = q1;
            row[2] = q2;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Output layer of the cost-to-go network (utils/pytorch_models.py:83-86: fc_out, 1000 -> out_dim, out_dim = 1 for every
// environment of the reference): out[r][o] = x[r] . w[o] + b[o].  A [m, 1024] x [1024, 1] product is a streaming pass
// over x (HBM-bound: 4 KB per row), not a GEMM — and the library GEMV it used to be picks its kernel, and with it the
// order of the 1024 additions, from m.  Here the order is FIXED: one wave per row, lane l accumulates the 4-element
// chunks l, l + 64, l + 128 ... left to right, then a xor-butterfly folds the 64 lane sums — a row's value is the same
// bits whatever row index, batch size or launch it is evaluated in, which is what lets the dedup-first search (kept
// children, packed) and the reference-order search (all children) be compared bit for bit.  The sums run in FLOAT64
// (products of two fp32 values are exact there; the kernel is HBM-bound, the fp64 FMAs are free) and are rounded to fp32
// once: at trained magnitudes (|h| ~ 25, one fp32 ulp = 1.9e-6) an fp32 summation alone spends a third of the north
// star's 1e-5 budget (measured: 1.08e-5 end to end with an fp32 chain, test_heuristic_tolerance_at_trained_network_magnitudes).
// x: fp32, or bf16 / fp16 (the non-parity modes' residual stream), converted on load.
// ---------------------------------------------------------------------------------------------------------------------
template <typename XT>
__device__ __forceinline__ float4 head_load4(const XT* p);
template <>
__device__ __forceinline__ float4 head_load4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <>
__device__ __forceinline__ float4 head_load4<_Float16>(const _Float16* p) {
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    const h4 v = *reinterpret_cast<const h4*>(p);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
template <>
__device__ __forceinline__ float4 head_load4<uint16_t>(const uint16_t* p) {  // bf16 bit patterns
    const uint2 v = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xFFFF0000u), __uint_as_float(v.y << 16),
                       __uint_as_float(v.y & 0xFFFF0000u));
}

constexpr int kHeadMaxOut = 8;
template <typename XT>
__global__ __launch_bounds__(256) void k_head_gemv(const XT* __restrict__ x, int64_t m, int k, int64_t ldx,
                                                   const float* __restrict__ w /*[n_out, k]*/, const float* __restrict__ b,
                                                   int n_out, float* __restrict__ out /*[m, n_out]*/) {
    extern __shared__ __attribute__((aligned(16))) uint8_t head_lds[];
    float* lw = reinterpret_cast<float*>(head_lds);  // the weights, once per workgroup
    for (int i = threadIdx.x; i < n_out * k; i += 256) lw[i] = w[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nch = k >> 2;
    for (int64_t r = (int64_t)blockIdx.x * 4 + wv; r < m; r += (int64_t)gridDim.x * 4) {
        const XT* row = x + r * ldx;
        double acc[kHeadMaxOut];
#pragma unroll
        for (int o = 0; o < kHeadMaxOut; o++) acc[o] = 0.0;
        for (int c = lane; c < nch; c += 64) {
            const float4 v = head_load4<XT>(row + 4 * c);
#pragma unroll
            for (int o = 0; o < kHeadMaxOut; o++) {
                if (o < n_out) {
                    const float4 q = *reinterpret_cast<const float4*>(lw + o * k + 4 * c);
                    acc[o] = fma((double)v.x, (double)q.x, acc[o]);
                    acc[o] = fma((double)v.y, (double)q.y, acc[o]);
                    acc[o] = fma((double)v.z, (double)q.z, acc[o]);
                    acc[o] = fma((double)v.w, (double)q.w, acc[o]);
                }
            }
        }
#pragma unroll
        for (int o = 0; o < kHeadMaxOut; o++) {
            if (o < n_out) {
                double s = acc[o];
                for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d);  // every lane ends with the same bits
                if (lane == 0) out[r * n_out + o] = (float)(s + (b ? (double)b[o] : 0.0));
            }
        }
    }
}

}  // namespace dca


using namespace dca;

extern "C" {

int dca_l1_supported(int state_dim, int depth) {
    return (state_dim == 54 && depth == 6) || (state_dim == depth && (depth == 16 || depth == 25 || depth == 36 || depth == 49));
}

int64_t dca_l1_kpad(int state_dim, int depth) { return (((int64_t)state_dim * depth + 15) / 16) * 16; }

int dca_l1_onehot_gemm(const uint8_t* nnet_in, int64_t m, int state_dim, int depth, const void* w_tiles, int planes,
                       int64_t n_pad, const float* bias, int relu, void* out, int out_dtype, int* overflow, void* stream) {
    DCA_ARG(nnet_in && w_tiles && bias && out && m >= 0 && planes >= 1 && planes <= 3 && n_pad >= 64 && n_pad % 64 == 0);
    DCA_ARG(out_dtype >= DCA_DT_F32 && out_dtype <= DCA_DT_E4M3);
    if (!dca_l1_supported(state_dim, depth)) {
        set_error("dca_l1_onehot_gemm: geometry (%d, %d) not instantiated (weight tile must fit LDS)", state_dim, depth);
        return DCA_E_BADARG;
    }
    if (m == 0) return 0;
    const uint8_t* wt = reinterpret_cast<const uint8_t*>(w_tiles);
    hipStream_t s = (hipStream_t)stream;
    switch (state_dim) {
        case 54: return launch_l1<54, 6>(planes, nnet_in, m, wt, bias, relu, out, out_dtype, n_pad, overflow, s);
        case 16: return launch_l1<16, 16>(planes, nnet_in, m, wt, bias, relu, out, out_dtype, n_pad, overflow, s);
        case 25: return launch_l1<25, 25>(planes, nnet_in, m, wt, bias, relu, out, out_dtype, n_pad, overflow, s);
        case 36: return launch_l1<36, 36>(planes, nnet_in, m, wt, bias, relu, out, out_dtype, n_pad, overflow, s);
        default: return launch_l1<49, 49>(planes, nnet_in, m, wt, bias, relu, out, out_dtype, n_pad, overflow, s);
    }
}

int dca_l1_onehot_gemm_mx(const uint8_t* nnet_in, int64_t m, int state_dim, int depth, const void* w_tiles, int64_t n_pad,
                          const float* bias, int relu, void* out8, void* out_scale, int64_t ld_sc, void* stream) {
    DCA_ARG(nnet_in && w_tiles && bias && out8 && out_scale && m >= 0 && n_pad >= 64 && n_pad % 64 == 0 && ld_sc >= n_pad / 64);
    if (!dca_l1_supported(state_dim, depth)) {
        set_error("dca_l1_onehot_gemm_mx: geometry (%d, %d) not instantiated", state_dim, depth);
        return DCA_E_BADARG;
    }
    if (m == 0) return 0;
    const uint8_t* wt = reinterpret_cast<const uint8_t*>(w_tiles);
    uint8_t* sc = reinterpret_cast<uint8_t*>(out_scale);
    hipStream_t s = (hipStream_t)stream;
    switch (state_dim) {
        case 54: return launch_l1<54, 6>(1, nnet_in, m, wt, bias, relu, out8, DCA_DT_E4M3, n_pad, nullptr, s, sc, ld_sc);
        case 16: return launch_l1<16, 16>(1, nnet_in, m, wt, bias, relu, out8, DCA_DT_E4M3, n_pad, nullptr, s, sc, ld_sc);
        case 25: return launch_l1<25, 25>(1, nnet_in, m, wt, bias, relu, out8, DCA_DT_E4M3, n_pad, nullptr, s, sc, ld_sc);
        case 36: return launch_l1<36, 36>(1, nnet_in, m, wt, bias, relu, out8, DCA_DT_E4M3, n_pad, nullptr, s, sc, ld_sc);
        default: return launch_l1<49, 49>(1, nnet_in, m, wt, bias, relu, out8, DCA_DT_E4M3, n_pad, nullptr, s, sc, ld_sc);
    }
}

int dca_act_split(const float* y, const float* bias, const float* skip, const float* col_scale, double alpha, int relu,
                  int64_t m, int64_t n, float* x_out, void* a3, int a3_planes, int* overflow, void* stream) {
    DCA_ARG(y && (a3 || x_out) && m >= 0 && n >= 4 && n % 4 == 0 && m * n < (1ll << 40));
    if (m == 0) return 0;
    const unsigned gx = (unsigned)((n + 255) / 256);
    int64_t gy = 4096 / gx;
    if (gy > (m + 15) / 16) gy = (m + 15) / 16;
    if (gy < 1) gy = 1;
    hipLaunchKernelGGL(k_act_split, dim3(gx, (unsigned)gy), dim3(256), 0, (hipStream_t)stream, y, bias, skip, col_scale,
                       (float)alpha, relu,
                       m, n, x_out, reinterpret_cast<_Float16*>(a3), a3_planes, overflow);
    return launch_check("k_act_split");
}

int dca_head_gemv(const void* x, int x_dtype, int64_t m, int k, int64_t ldx, const float* w, const float* bias, int n_out,
                  float* out, void* stream) {
    DCA_ARG(x && w && out && m >= 0 && k >= 4 && k % 4 == 0 && ldx >= k && n_out >= 1 && n_out <= kHeadMaxOut);
    DCA_ARG(x_dtype == DCA_DT_F32 || x_dtype == DCA_DT_F16 || x_dtype == DCA_DT_BF16);
    DCA_ARG((size_t)n_out * (size_t)k * sizeof(float) <= 64 * 1024);
    DCA_ARG((uintptr_t)x % (x_dtype == DCA_DT_F32 ? 16 : 8) == 0 && ldx % 4 == 0 && (uintptr_t)w % 16 == 0);
    if (m == 0) return 0;
    int64_t blocks = (m + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    const size_t lds = (size_t)n_out * (size_t)k * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    if (x_dtype == DCA_DT_F32)
        hipLaunchKernelGGL(k_head_gemv<float>, dim3((unsigned)blocks), dim3(256), lds, s, reinterpret_cast<const float*>(x), m, k,
                           ldx, w, bias, n_out, out);
    else if (x_dtype == DCA_DT_F16)
        hipLaunchKernelGGL(k_head_gemv<_Float16>, dim3((unsigned)blocks), dim3(256), lds, s,
                           reinterpret_cast<const _Float16*>(x), m, k, ldx, w, bias, n_out, out);
    else
        hipLaunchKernelGGL(k_head_gemv<uint16_t>, dim3((unsigned)blocks), dim3(256), lds, s,
                           reinterpret_cast<const uint16_t*>(x), m, k, ldx, w, bias, n_out, out);
    return launch_check("k_head_gemv");
}

}  // extern "C"
