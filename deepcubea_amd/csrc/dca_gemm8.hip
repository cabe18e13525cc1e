// dca_gemm8.hip — the dense layers of the cost-to-go network in the NON-parity fp8 mode (`--nnet_dtype fp8`; SURVEY §8(f)-2,
// "a hand-written bf16 (then fp8) GEMM with the bias/skip/ReLU epilogue"): OCP e4m3 operands, fp32
// accumulation on v_mfma_f32_32x32x64_f8f6f4 (the non-scaled form of gfx950's block-scaled MFMA: twice the bf16 rate),
// ONE launch per layer with the whole tail — dequantisation, bias, residual add, ReLU, and the QUANTISATION of the result for
// the next layer — in the epilogue.
//
// Reference arithmetic (utils/pytorch_models.py:57-86, BatchNorm folded):  v = relu?(x . W^T + b (+ skip)), here with
//   x ~ s_x * x8 (one scale per activation tensor, calibrated by the caller), W[n,:] ~ s_w[n] * w8[n,:] (one scale per output
//   unit), so v = (x8 . w8^T)[m,n] * scale[n] + b[n] (+ skip), scale[n] = s_x * s_w[n] handed in by the caller.
// The residual stream stays bf16 (out16 / skip); the next layer's operand leaves as e4m3(sat(v * out8_scale)).
//
// Tile and schedule are those of csrc/dca_gemm16.hip variant 2 (the derivation and the RAW / WAR argument are written out
// there): 256 x 256 outputs per workgroup, 8 waves as 2 (M) x 4 (N), the two wave rows one barrier apart (one multiplies while
// the other reads fragments and issues DMA), a K-tile of 128 BYTES per row — the same 128-byte LDS rows, the same XOR swizzle
// on the DMA source address, the same four 16 KB half-tile slots per K-tile and the same counted s_waitcnt vmcnt(10) — only a
// K-tile is now 128 deep: per phase a wave issues 4 MFMAs of 32x32x64 (16 passes each) where the bf16 kernel issues 8 of
// 32x32x16 (8 passes), i.e. the same matrix-pipe time for twice the products.  A lane's 32-byte MFMA operand is two 16-byte
// chunks of its row (lane half h of K-step t takes chunks 4t + 2h and 4t + 2h + 1); A and B use the same chunk -> byte
// order, which is all the instruction needs (it pairs byte b of the A lane with byte b of the B lane of the same half).
#include <atomic>
#include <type_traits>

#include "dca_common.h"

namespace dca {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int EBM = 256, EBN = 256, EBK = 128, ETHREADS = 512;
constexpr int ESLOT = 128 * 128;  // one half-tile: 128 rows x 128 B
constexpr int EBUF = 4 * ESLOT;   // one K-tile: A01 | A23 | B0 | B1
constexpr int ELDS = 2 * EBUF;    // 128 KB
constexpr int ES_A01 = 0, ES_A23 = 1, ES_B0 = 2, ES_B1 = 3;

struct Gemm8Args {
    const uint8_t* a;    // [m, lda] e4m3
    const uint8_t* w;    // [n, ldw] e4m3 (row = output unit)
    const float* scale;  // [n]: activation scale x weight scale of the unit
    const float* bias;   // [n] or null
    const uint16_t* skip;  // [m, ldo16] bf16 or null
    uint16_t* out16;     // [m, ldo16] bf16 or null
    uint8_t* out8;       // [m, ldo8] e4m3 or null
    float out8_scale;    // out8 = e4m3(sat(v * out8_scale)) — unless block scales are written (out8_scale below)
    int relu;
    int64_t m;
    int n, k;
    int64_t lda, ldw, ldo16, ldo8;
    // block-scaled ("MX") operands and results: one E8M0 byte (value 2^(byte - 127)) per 64 consecutive K elements of a row
    const uint8_t* a_scale;  // [m, ld_asc] (k / 64 bytes per row) or null: per-tensor scaling (scale[] carries it)
    int64_t ld_asc;
    uint8_t* out8_sc;        // [m, ld_osc] (n / 64 bytes per row) or null: out8 = e4m3(sat(v * out8_scale)) with the static scalar
    int64_t ld_osc;
    int out8_scale_ptr_set;  // (out8_sc != null: a uniform scalar the epilogue branches on)
};

// Largest value over the 16 lanes of a DPP row (lanes 16 r .. 16 r + 15), in every lane of the row: four rotate-and-max steps on
// the VALU (row_ror 8, 4, 2, 1).  (A __shfl_xor chain is four dependent ds_bpermute round trips through the LDS crossbar: 8 us
// per tile in the layer tail, measured — the whole cost of writing block scales.)
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, false)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xF, 0xF, false)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xF, 0xF, false)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xF, 0xF, false)));
    return v;
}

// E8M0 block scale of a group whose largest magnitude is `amax`: the smallest power of two 2^e with amax * 2^-e <= 448 (the
// largest e4m3 value), as (byte = e + 127, multiplier 2^-e).  amax = 0 (or a tiny group) takes the smallest scale.
__device__ __forceinline__ uint32_t e8m0_of_amax(float amax, float& inv) {
    const float t = amax * (1.0f / 448.0f);
    const uint32_t u = __float_as_uint(t);
    int e = (int)((u >> 23) & 0xFFu) - 127 + ((u & 0x7FFFFFu) ? 1 : 0);  // ceil(log2 t) for normal t
    e = e < -126 ? -126 : (e > 126 ? 126 : e);
    inv = __uint_as_float((uint32_t)(127 - e) << 23);  // 2^-e
    return (uint32_t)(e + 127);
}

__device__ __forceinline__ uint32_t swz128b(uint32_t row, uint32_t chunk) { return row * 128u + ((chunk ^ ((row >> 1) & 7u)) << 4); }

typedef __bf16 g8_bf16x2 __attribute__((ext_vector_type(2)));
typedef float g8_f32x2 __attribute__((ext_vector_type(2)));
// round to nearest even, NaN stays NaN: gfx950's v_cvt_pk_bf16_f32, two values per instruction (the integer sequence it
// replaces took six per value — a visible share of the layer tail, csrc/dca_gemm16.hip)
__device__ __forceinline__ uint32_t f32_to_bf16x2(float lo, float hi) {
    const g8_f32x2 v = {lo, hi};
    const g8_bf16x2 b = __builtin_convertvector(v, g8_bf16x2);
    uint32_t u;
    __builtin_memcpy(&u, &b, 4);
    return u;
}
__device__ __forceinline__ uint16_t f32_to_bf16(float f) { return (uint16_t)(f32_to_bf16x2(f, 0.f) & 0xFFFFu); }
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }

// four floats -> four e4m3 bytes (round to nearest even, saturating at +-448: e4m3fn has no infinity)
__device__ __forceinline__ uint32_t pack_e4m3(float a, float b, float c, float d) {
    a = fminf(fmaxf(a, -448.f), 448.f);
    b = fminf(fmaxf(b, -448.f), 448.f);
    c = fminf(fmaxf(c, -448.f), 448.f);
    d = fminf(fmaxf(d, -448.f), 448.f);
    uint32_t r = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    r = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(c, d, (int)r, true);
    return r;
}

#define DCA_BAR() asm volatile("s_barrier" ::: "memory")
#define DCA_RD_DONE_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define DCA_VMCNT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")

// Layer tail.  MODE < 0: any shape, any form, block-scaled results included, every option tested at run time.  MODE >= 0 (bit 0
// skip, bit 1 bf16 output, bit 2 e4m3 output with the static scale, bit 3 bias): the same tail compiled for ONE of the network's
// layer forms on a tile inside the matrix, with ReLU — no per-row / per-element tests left (round 5; the 1024-wide e4m3 layers
// spend 40-50 % of their time in this tail, csrc/dca_gemm16.hip has the measurements that led here).
template <int MODE>
__device__ __forceinline__ void gemm8_tail_as(const Gemm8Args& p, uint8_t* lds, const f32x16 (&acc)[4][2], int64_t m0, int n0, int w, int wm,
                                              int wn, int lane, int l31, int h) {
    constexpr bool G = MODE < 0;
    const bool has_skip = G ? p.skip != nullptr : (MODE & 1) != 0;
    const bool has16 = G ? p.out16 != nullptr : (MODE & 2) != 0;
    const bool has8 = G ? p.out8 != nullptr : (MODE & 4) != 0;
    const bool has_bias = G ? p.bias != nullptr : (MODE & 8) != 0;
    const bool relu = G ? p.relu != 0 : true;
    // Accumulator layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).  Each wave
    // transposes its tile through its own 16 KB of the (now idle) LDS, 32 rows at a time: a lane then owns 4 consecutive
    // columns of a row — one 8-byte skip load, one 8-byte bf16 store, one 4-byte e4m3 store.
    float* sl = reinterpret_cast<float*>(lds + w * 16384);
    float sc[2], bv[2];
#pragma unroll
    for (int jn = 0; jn < 2; jn++) {
        const int col = n0 + wn * 64 + jn * 32 + l31;
        sc[jn] = (G ? col < p.n : true) ? p.scale[col] : 0.f;
        bv[jn] = ((G ? col < p.n : true) && has_bias) ? p.bias[col] : 0.f;
    }
    const int c4 = (lane & 15) * 4;  // this lane's 4 columns inside the wave's 64
    const int colg = n0 + wn * 64 + c4;
    const bool full4 = G ? colg + 3 < p.n : true;
    // residual rows: the loads of round i+1 are issued before round i is worked on (two register sets), so only the first
    // round waits a full memory latency — the waits between rounds would otherwise add up (4 x ~1.5 us per tile)
    uint2 sk[2][8];
    auto load_skip = [&](int i, uint2 (&dst)[8]) {
        const int64_t rb = m0 + wm * 128 + i * 32;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int64_t r = rb + q * 4 + (lane >> 4);
            dst[q] = make_uint2(0u, 0u);
            if (has_skip && (G ? (r < p.m && full4) : true)) dst[q] = *reinterpret_cast<const uint2*>(p.skip + r * p.ldo16 + colg);
        }
    };
    if (has_skip) load_skip(0, sk[0]);
#pragma unroll
    for (int i = 0; i < 4; i++) {
#pragma unroll
        for (int jn = 0; jn < 2; jn++)
#pragma unroll
            for (int reg = 0; reg < 16; reg++)
                sl[((reg & 3) + 8 * (reg >> 2) + 4 * h) * 64 + jn * 32 + l31] = acc[i][jn][reg] * sc[jn] + bv[jn];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private slice: no barrier, just the wave's own writes
        const int64_t rbase = m0 + wm * 128 + i * 32;
        if (has_skip && i + 1 < 4) load_skip(i + 1, sk[(i + 1) & 1]);
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int rl = q * 4 + (lane >> 4);
            const int64_t r = rbase + rl;
            const float4 v = *reinterpret_cast<const float4*>(sl + rl * 64 + c4);
            if (G && r >= p.m) continue;
            float u[4] = {v.x, v.y, v.z, v.w};
            if (full4) {
                if (has_skip) {
                    const uint2 s2 = sk[i & 1][q];
                    u[0] += bf16_to_f32((uint16_t)(s2.x & 0xFFFFu));
                    u[1] += bf16_to_f32((uint16_t)(s2.x >> 16));
                    u[2] += bf16_to_f32((uint16_t)(s2.y & 0xFFFFu));
                    u[3] += bf16_to_f32((uint16_t)(s2.y >> 16));
                }
                if (relu) {
#pragma unroll
                    for (int e = 0; e < 4; e++) u[e] = fmaxf(u[e], 0.f);
                }
                if (has16) {
                    uint2 ov;
                    ov.x = f32_to_bf16x2(u[0], u[1]);
                    ov.y = f32_to_bf16x2(u[2], u[3]);
                    *reinterpret_cast<uint2*>(p.out16 + r * p.ldo16 + colg) = ov;
                }
                if (has8) {
                    float s = p.out8_scale;
                    if (G && p.out8_scale_ptr_set) {
                        // block scale of this row's 64 columns (the wave's slice: 16 lanes x 4 columns): largest magnitude over the
                        // 16 lanes, the power of two that brings it into e4m3's range, one byte per (row, 64 columns)
                        const float am = row16_max(fmaxf(fmaxf(fabsf(u[0]), fabsf(u[1])), fmaxf(fabsf(u[2]), fabsf(u[3]))));
                        const uint32_t sb = e8m0_of_amax(am, s);
                        if ((lane & 15) == 0) p.out8_sc[r * p.ld_osc + (colg >> 6)] = (uint8_t)sb;
                    }
                    *reinterpret_cast<uint32_t*>(p.out8 + r * p.ldo8 + colg) = pack_e4m3(u[0] * s, u[1] * s, u[2] * s, u[3] * s);
                }
            } else {  // ragged right edge: element-wise
                for (int e = 0; e < 4 && colg + e < p.n; e++) {
                    float ue = u[e] + (p.skip ? bf16_to_f32(p.skip[r * p.ldo16 + colg + e]) : 0.f);
                    if (p.relu) ue = fmaxf(ue, 0.f);
                    if (p.out16) p.out16[r * p.ldo16 + colg + e] = f32_to_bf16(ue);
                    if (p.out8) p.out8[r * p.ldo8 + colg + e] = (uint8_t)(pack_e4m3(ue * p.out8_scale, 0.f, 0.f, 0.f) & 0xFFu);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the slice is rewritten by the next 32 rows
    }
}

// MXIN: the A operand carries E8M0 block scales (one per 64 K elements of a row).  The whole K range of the tile's 256 rows
// (k / 64 bytes a row: 4 KB at k = 1024, 20 KB at 5120) is staged into LDS behind the operand slots in the prologue — ordinary
// loads, before the K loop, so the loop's counted LDS-DMA waits are untouched — and a lane fetches the two scales of its row
// for a K-tile with one ds_read_u16.  gfx950's scaled MFMA multiplies a lane's 32 products by 2^(sa - 127) * 2^(sb - 127),
// sa / sb being a byte of the lane's scale registers: both lane halves of a row pass the row's scale of the 64-deep step,
// the weights' side passes 127 (their per-output-unit fp32 scales stay in the epilogue).
template <bool MXIN>
__global__ __launch_bounds__(ETHREADS, 2) void k_gemm8(const Gemm8Args p) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, l31 = lane & 31, h = lane >> 5;
    const int wm = w >> 2, wn = w & 3;
    const int nNt = (p.n + EBN - 1) / EBN;
    const int64_t nMt = (p.m + EBM - 1) / EBM;
    const int64_t bid = blockIdx.x;
    const int64_t slot = bid >> 3;
    const int64_t mt = (slot / nNt) * 8 + (bid & 7);  // the N tiles of one M tile sit on one XCD (workgroup b runs on XCD b % 8)
    const int nt = (int)(slot % nNt);
    if (mt >= nMt) return;
    const int64_t m0 = mt * EBM;
    const int n0 = nt * EBN;

    // DMA map: instruction q (0, 1) of wave w fills local rows [(q*8 + w)*8, +8) of a half-tile slot; lane i lands on local
    // row r = that + (i >> 3), physical chunk i & 7, and fetches logical chunk (i & 7) ^ ((r >> 1) & 7) of the matrix row
    // the slot's local row r stands for.  Rows past the matrix edge are clamped: their products are never stored.
    const uint8_t* src[4][2];
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const uint32_t r = (uint32_t)((q * 8 + w) * 8 + (lane >> 3));
            const uint32_t c = (uint32_t)(lane & 7) ^ ((r >> 1) & 7u);
            if (u < 2) {
                int64_t gr = m0 + (r >> 6) * 128 + (u == ES_A23 ? 64 : 0) + (r & 63);
                gr = gr < p.m ? gr : p.m - 1;
                src[u][q] = p.a + gr * p.lda + c * 16;
            } else {
                int gn = n0 + (int)((r >> 5) * 64 + (u == ES_B1 ? 32 : 0) + (r & 31));
                gn = gn < p.n ? gn : p.n - 1;
                src[u][q] = p.w + (int64_t)gn * p.ldw + c * 16;
            }
        }
    auto issue = [&](int u, int buf, int k0) {
#pragma unroll
        for (int q = 0; q < 2; q++) {
            uint8_t* dst = lds + buf * EBUF + u * ESLOT + (q * 8 + w) * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[u][q] + k0),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int jn = 0; jn < 2; jn++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][jn][e] = 0.f;

    // fragment addresses inside a slot: local row = (wave's block) * 32 + l31; K-step ks of the tile, half j of the operand
    uint32_t foff[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ks++)
#pragma unroll
        for (int j = 0; j < 2; j++) foff[ks][j] = swz128b((uint32_t)l31, 4u * ks + 2u * (uint32_t)h + j);
    const uint32_t a_row0 = (uint32_t)wm * 64u * 128u;  // A slots: this wave row's 64 local rows
    const uint32_t b_row0 = (uint32_t)wn * 32u * 128u;  // B slots: this wave column's 32 local rows

    // (typed vector loads: see dca_gemm16.hip — a struct-typed load would be ordered behind the LDS-DMA in flight)
    auto frag = [&](const uint8_t* q, int ks) {
        const i32x4 lo = *reinterpret_cast<const i32x4*>(q + foff[ks][0]);
        const i32x4 hi = *reinterpret_cast<const i32x4*>(q + foff[ks][1]);
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    i32x8 av[2][2], wv0[2], wv1[2];
    // block scales of the A rows in `av`: bytes 0 / 1 = the two 64-deep steps of the current K-tile
    uint32_t sa[2] = {0u, 0u};
    const uint32_t SK = MXIN ? (uint32_t)(p.k / 64) : 0u;
    uint8_t* lsc = lds + ELDS;
    const uint32_t sc_row0 = ((uint32_t)wm * 128u + (uint32_t)l31) * SK;
    auto read_a = [&](const uint8_t* base, int u, int kt) {
#pragma unroll
        for (int ii = 0; ii < 2; ii++) {
#pragma unroll
            for (int ks = 0; ks < 2; ks++) av[ii][ks] = frag(base + u * ESLOT + a_row0 + ii * 4096, ks);
            if constexpr (MXIN)  // tile row = wm * 128 + (u == A23 ? 64 : 0) + ii * 32 + l31
                sa[ii] = *reinterpret_cast<const uint16_t*>(lsc + sc_row0 + (uint32_t)((u == ES_A23 ? 64 : 0) + ii * 32) * SK + 2u * (uint32_t)kt);
        }
    };
    auto read_b = [&](const uint8_t* base, int u, i32x8 (&wv)[2]) {
#pragma unroll
        for (int ks = 0; ks < 2; ks++) wv[ks] = frag(base + u * ESLOT + b_row0, ks);
    };
#define DCA_MMA4_STEP(I0, JN, WV, KS)                                                                                    \
    _Pragma("unroll") for (int ii = 0; ii < 2; ii++) {                                                                   \
        if constexpr (MXIN) /* op_sel = KS: byte KS of the lane's scale register (the K-tile's two 64-deep steps) */     \
            acc[(I0) + ii][JN] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av[ii][KS], WV[KS], acc[(I0) + ii][JN], \
                                                                                 0, 0, KS, (int)sa[ii], 0, 127);         \
        else                                                                                                             \
            acc[(I0) + ii][JN] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av[ii][KS], WV[KS], acc[(I0) + ii][JN], \
                                                                                 0, 0, 0, 0, 0, 0);                      \
    }
#define DCA_MMA4(I0, JN, WV)                                                                                             \
    do {                                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        __builtin_amdgcn_s_setprio(1);                                                                                   \
        DCA_MMA4_STEP(I0, JN, WV, 0)                                                                                     \
        DCA_MMA4_STEP(I0, JN, WV, 1)                                                                                     \
        __builtin_amdgcn_s_setprio(0);                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
    } while (0)

    const int nk = p.k / EBK;
    // one K-tile; N1 / N2: tiles kt+1 / kt+2 exist (compile-time: the steady-state body is branch-free).  On entry: issued =
    // all of tile kt and A01, B0, B1 of kt+1; landed and visible = A01, B0 of kt.  The vmcnt numbers count the DMA
    // instructions (2 per half-tile) issued AFTER the half-tile being waited for.
    auto tile = [&](int kt, auto n1c, auto n2c) {
        constexpr bool N1 = decltype(n1c)::value, N2 = decltype(n2c)::value;
        const int b = kt & 1;
        const uint8_t* base = lds + b * EBUF;
        // phase 1: (A01, B0); restage A23 of kt+1; retire B1 of kt
        read_b(base, ES_B0, wv0);
        read_a(base, ES_A01, kt);
        if constexpr (N1) {
            issue(ES_A23, b ^ 1, (kt + 1) * EBK);
            DCA_VMCNT(10);
        } else {
            DCA_VMCNT(2);
        }
        DCA_RD_DONE_BAR();
        DCA_MMA4(0, 0, wv0);
        DCA_BAR();
        // phase 2: (A01, B1); restage A01 of kt+2; retire A23 of kt
        read_b(base, ES_B1, wv1);
        if constexpr (N2) {
            issue(ES_A01, b, (kt + 2) * EBK);
            DCA_VMCNT(10);
        } else if constexpr (N1) {
            DCA_VMCNT(8);
        } else {
            DCA_VMCNT(0);
        }
        DCA_RD_DONE_BAR();
        DCA_MMA4(0, 1, wv1);
        DCA_BAR();
        // phase 3: (A23, B1); restage B0 of kt+2
        read_a(base, ES_A23, kt);
        if constexpr (N2) issue(ES_B0, b, (kt + 2) * EBK);
        DCA_RD_DONE_BAR();
        DCA_MMA4(2, 1, wv1);
        DCA_BAR();
        // phase 4: (A23, B0) from registers; restage B1 of kt+2; retire A01, B0 of kt+1
        if constexpr (N2) {
            issue(ES_B1, b, (kt + 2) * EBK);
            DCA_VMCNT(10);
        } else if constexpr (N1) {
            DCA_VMCNT(4);
        }
        DCA_RD_DONE_BAR();
        DCA_MMA4(2, 0, wv0);
        DCA_BAR();
    };

    // the tile's block scales: 256 rows x SK bytes (rows past m: clamped), requested BEFORE the operand DMAs — the oldest
    // entries of the memory queue — so that their round trip runs under the DMAs' and a counted wait retires them alone
    // (The loads and the LDS writes of the scale words are inline asm: for a load it knows about, hipcc waits vmcnt(0) at the
    // first use of its result while LDS-DMA is in flight — the whole prologue would drain — so the queue is counted by hand.)
    constexpr int kScW = 16;  // words per thread at most: 256 rows x 128 scale bytes (k = 8192) / 4 / 512 threads
    uint32_t scw[kScW];  // (prologue only: dead before the accumulators and operand registers are live)
    if constexpr (MXIN) {
        const uint32_t words = 256u * SK / 4u;
#pragma unroll
        for (int j = 0; j < kScW; j++) {
            if ((uint32_t)j * ETHREADS >= words) continue;  // (uniform: k = 1024 takes 2 passes, k = 5120 takes 10)
            const uint32_t q = (uint32_t)t + (uint32_t)j * ETHREADS, qc = q < words ? q : 0u;
            const uint32_t row = (qc * 4u) / SK, col = (qc * 4u) - row * SK;
            int64_t gr = m0 + row;
            gr = gr < p.m ? gr : p.m - 1;
            const uint8_t* src_sc = p.a_scale + gr * p.ld_asc + col;
            asm volatile("global_load_dword %0, %1, off" : "=v"(scw[j]) : "v"(src_sc) : "memory");
        }
    }
    issue(ES_A01, 0, 0);
    issue(ES_B0, 0, 0);
    issue(ES_B1, 0, 0);
    issue(ES_A23, 0, 0);
    if (nk > 1) {
        issue(ES_A01, 1, EBK);
        issue(ES_B0, 1, EBK);
        issue(ES_B1, 1, EBK);
    }
    if constexpr (MXIN) {
        const uint32_t words = 256u * SK / 4u;
        if (nk > 1)
            DCA_VMCNT(14);  // the scale words (issued first) are here; the 14 DMA instructions behind them stay in flight
        else
            DCA_VMCNT(8);
        const uint32_t lsc_off = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) uint8_t*)lsc);
#pragma unroll
        for (int j = 0; j < kScW; j++) {
            if ((uint32_t)j * ETHREADS >= words) continue;
            const uint32_t q = (uint32_t)t + (uint32_t)j * ETHREADS;
            const uint32_t dst_off = lsc_off + (q < words ? q : (uint32_t)t) * 4u;  // (surplus lanes rewrite a word of their own: same value)
            asm volatile("ds_write_b32 %0, %1" ::"v"(dst_off), "v"(q < words ? scw[j] : scw[0]) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the LDS image is written before the barrier below publishes it
    }
    if (nk > 1) {
        DCA_VMCNT(10);  // A01, B0 of tile 0 have landed
    } else {
        DCA_VMCNT(4);
    }
    DCA_BAR();
    if (wm == 1) DCA_BAR();  // the second wave row runs one barrier behind the first from here on
    {
        int kt = 0;
        for (; kt + 2 < nk; kt++) tile(kt, std::true_type{}, std::true_type{});
        if (kt + 1 < nk) {
            tile(kt, std::true_type{}, std::false_type{});
            kt++;
        }
        tile(kt, std::false_type{}, std::false_type{});
    }
    if (wm == 0) DCA_BAR();  // ... and the first waits for it here
#undef DCA_MMA4
#undef DCA_MMA4_STEP

    // the layer tail: the network's own forms on a tile inside the matrix (uniform over the workgroup) take a tail compiled for
    // them; everything else — ragged edges, block-scaled results, no ReLU — the general one
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // every wave is done with the operand slots
    {
        const bool inside = m0 + EBM <= p.m && n0 + EBN <= p.n && p.relu && !p.out8_scale_ptr_set;
        const int mode = inside ? ((p.skip ? 1 : 0) | (p.out16 ? 2 : 0) | (p.out8 ? 4 : 0) | (p.bias ? 8 : 0)) : -1;
        switch (mode) {
            case 12: gemm8_tail_as<12>(p, lds, acc, m0, n0, w, wm, wn, lane, l31, h); break;  // bias + ReLU -> e4m3
            case 14: gemm8_tail_as<14>(p, lds, acc, m0, n0, w, wm, wn, lane, l31, h); break;  // bias + ReLU -> bf16 and e4m3 (the 5120 -> 1024 layer)
            case 15: gemm8_tail_as<15>(p, lds, acc, m0, n0, w, wm, wn, lane, l31, h); break;  // bias + skip + ReLU -> bf16 and e4m3
            case 11: gemm8_tail_as<11>(p, lds, acc, m0, n0, w, wm, wn, lane, l31, h); break;  // last block: bias + skip + ReLU -> bf16
            case 7: gemm8_tail_as<7>(p, lds, acc, m0, n0, w, wm, wn, lane, l31, h); break;    // skip + ReLU -> bf16 and e4m3 (no bias)
            case 3: gemm8_tail_as<3>(p, lds, acc, m0, n0, w, wm, wn, lane, l31, h); break;    // skip + ReLU -> bf16 (no bias)
            default: gemm8_tail_as<-1>(p, lds, acc, m0, n0, w, wm, wn, lane, l31, h); break;
        }
    }
}
#undef DCA_VMCNT
#undef DCA_RD_DONE_BAR
#undef DCA_BAR

// bf16 / fp32 [m, n] -> e4m3(sat(x * scale)): the entry into an fp8 layer for activations that did not come out of an fp8 epilogue
template <typename T>
__global__ __launch_bounds__(256) void k_quant_e4m3(const T* __restrict__ x, int64_t m, int64_t n, int64_t ld, float scale,
                                                    uint8_t* __restrict__ out, int64_t ldo) {
    const int64_t n4 = n / 4;
    const int64_t total = m * n4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / n4, c = (i - r * n4) * 4;
        float u[4];
        if constexpr (std::is_same<T, float>::value) {
            const float4 v = *reinterpret_cast<const float4*>(x + r * ld + c);
            u[0] = v.x, u[1] = v.y, u[2] = v.z, u[3] = v.w;
        } else {
            const uint2 v = *reinterpret_cast<const uint2*>(x + r * ld + c);
            u[0] = bf16_to_f32((uint16_t)(v.x & 0xFFFFu)), u[1] = bf16_to_f32((uint16_t)(v.x >> 16));
            u[2] = bf16_to_f32((uint16_t)(v.y & 0xFFFFu)), u[3] = bf16_to_f32((uint16_t)(v.y >> 16));
        }
        *reinterpret_cast<uint32_t*>(out + r * ldo + c) = pack_e4m3(u[0] * scale, u[1] * scale, u[2] * scale, u[3] * scale);
    }
}

}  // namespace dca

using namespace dca;

extern "C" {

int dca_gemm8(const void* a, int64_t m, int k, int64_t lda, const void* w, int n, int64_t ldw, const float* scale, const float* bias,
              const void* skip, int relu, void* out16, int64_t ldo16, void* out8, int64_t ldo8, double out8_scale, void* stream) {
    DCA_ARG(a && w && scale && (out16 || out8) && m >= 0 && n >= 1 && k >= EBK && k % EBK == 0);
    DCA_ARG(lda >= k && ldw >= k && lda % 16 == 0 && ldw % 16 == 0);
    DCA_ARG(((uintptr_t)a | (uintptr_t)w) % 16 == 0);
    DCA_ARG((!out16 && !skip) || (ldo16 >= n && ldo16 % 4 == 0 && ((uintptr_t)out16 | (uintptr_t)skip) % 8 == 0));
    DCA_ARG(!out8 || (ldo8 >= n && ldo8 % 4 == 0 && (uintptr_t)out8 % 4 == 0 && out8_scale > 0.0));
    {   // the A tiles are re-read through LDS-DMA while the epilogues of other tiles write: no output may overlap the operand
        const uintptr_t a0 = (uintptr_t)a, a1 = a0 + (m > 0 ? (size_t)(m - 1) * (size_t)lda + (size_t)k : 0);
        auto overlaps = [&](const void* o, int64_t ld, size_t esz) {
            if (!o || m == 0) return false;
            const uintptr_t o0 = (uintptr_t)o, o1 = o0 + ((size_t)(m - 1) * (size_t)ld + (size_t)n) * esz;
            return o0 < a1 && a0 < o1;
        };
        DCA_ARG(!overlaps(out8, ldo8, 1) && !overlaps(out16, ldo16, 2));
    }
    if (m == 0) return 0;
    {   // the dynamic-LDS limit is a per-device function attribute: set it once for every device this process uses
        static std::atomic<uint64_t> attr_devs{0};
        int dev = 0;
        DCA_HIP(hipGetDevice(&dev));
        const uint64_t bit = 1ull << (dev & 63);
        if (!(attr_devs.load(std::memory_order_acquire) & bit)) {
            DCA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm8<false>), hipFuncAttributeMaxDynamicSharedMemorySize, ELDS));
            attr_devs.fetch_or(bit, std::memory_order_release);
        }
    }
    Gemm8Args p;
    p.a = reinterpret_cast<const uint8_t*>(a);
    p.w = reinterpret_cast<const uint8_t*>(w);
    p.scale = scale;
    p.bias = bias;
    p.skip = reinterpret_cast<const uint16_t*>(skip);
    p.out16 = reinterpret_cast<uint16_t*>(out16);
    p.out8 = reinterpret_cast<uint8_t*>(out8);
    p.out8_scale = (float)out8_scale;
    p.relu = relu;
    p.m = m;
    p.n = n;
    p.k = k;
    p.lda = lda;
    p.ldw = ldw;
    p.ldo16 = ldo16;
    p.ldo8 = ldo8;
    const int64_t nMt = (m + EBM - 1) / EBM;
    const int64_t nNt = (n + EBN - 1) / EBN;
    const int64_t blocks = ((nMt + 7) / 8) * 8 * nNt;
    if (blocks > 0x7FFFFFFFll) {
        set_error("dca_gemm8: too many tiles");
        return DCA_E_BADARG;
    }
    p.a_scale = nullptr;
    p.ld_asc = 0;
    p.out8_sc = nullptr;
    p.ld_osc = 0;
    p.out8_scale_ptr_set = 0;
    hipLaunchKernelGGL(k_gemm8<false>, dim3((unsigned)blocks), dim3(ETHREADS), ELDS, (hipStream_t)stream, p);
    return launch_check("k_gemm8");
}

int dca_gemm8_mx(const void* a, const void* a_scale, int64_t m, int k, int64_t lda, int64_t ld_asc, const void* w, int n, int64_t ldw,
                 const float* w_scale, const float* bias, const void* skip, int relu, void* out16, int64_t ldo16, void* out8,
                 int64_t ldo8, void* out8_scale, int64_t ld_osc, void* stream) {
    DCA_ARG(a && a_scale && w && w_scale && (out16 || out8) && m >= 0 && n >= 1 && k >= 256 && k % 256 == 0 && k <= 8192);
    DCA_ARG(lda >= k && ldw >= k && lda % 16 == 0 && ldw % 16 == 0 && ld_asc >= k / 64 && ld_asc % 4 == 0);
    DCA_ARG(((uintptr_t)a | (uintptr_t)w) % 16 == 0 && (uintptr_t)a_scale % 4 == 0);
    DCA_ARG((!out16 && !skip) || (ldo16 >= n && ldo16 % 4 == 0 && ((uintptr_t)out16 | (uintptr_t)skip) % 8 == 0));
    DCA_ARG((out8 != nullptr) == (out8_scale != nullptr));
    DCA_ARG(!out8 || (ldo8 >= n && ldo8 % 4 == 0 && (uintptr_t)out8 % 4 == 0 && n % 64 == 0 && ld_osc >= n / 64));
    {
        const uintptr_t a0 = (uintptr_t)a, a1 = a0 + (m > 0 ? (size_t)(m - 1) * (size_t)lda + (size_t)k : 0);
        auto overlaps = [&](const void* o, int64_t ld, size_t esz) {
            if (!o || m == 0) return false;
            const uintptr_t o0 = (uintptr_t)o, o1 = o0 + ((size_t)(m - 1) * (size_t)ld + (size_t)n) * esz;
            return o0 < a1 && a0 < o1;
        };
        DCA_ARG(!overlaps(out8, ldo8, 1) && !overlaps(out16, ldo16, 2) && out8_scale != a_scale);
    }
    if (m == 0) return 0;
    const int lds_bytes = ELDS + 256 * (k / 64);
    {
        static std::atomic<uint64_t> attr_devs{0};
        int dev = 0;
        DCA_HIP(hipGetDevice(&dev));
        const uint64_t bit = 1ull << (dev & 63);
        if (!(attr_devs.load(std::memory_order_acquire) & bit)) {
            DCA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm8<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        ELDS + 256 * (8192 / 64)));
            attr_devs.fetch_or(bit, std::memory_order_release);
        }
    }
    Gemm8Args p;
    p.a = reinterpret_cast<const uint8_t*>(a);
    p.w = reinterpret_cast<const uint8_t*>(w);
    p.scale = w_scale;
    p.bias = bias;
    p.skip = reinterpret_cast<const uint16_t*>(skip);
    p.out16 = reinterpret_cast<uint16_t*>(out16);
    p.out8 = reinterpret_cast<uint8_t*>(out8);
    p.out8_scale = 1.f;
    p.relu = relu;
    p.m = m;
    p.n = n;
    p.k = k;
    p.lda = lda;
    p.ldw = ldw;
    p.ldo16 = ldo16;
    p.ldo8 = ldo8;
    p.a_scale = reinterpret_cast<const uint8_t*>(a_scale);
    p.ld_asc = ld_asc;
    p.out8_sc = reinterpret_cast<uint8_t*>(out8_scale);
    p.ld_osc = ld_osc;
    p.out8_scale_ptr_set = out8_scale != nullptr ? 1 : 0;
    const int64_t nMt = (m + EBM - 1) / EBM;
    const int64_t nNt = (n + EBN - 1) / EBN;
    const int64_t blocks = ((nMt + 7) / 8) * 8 * nNt;
    if (blocks > 0x7FFFFFFFll) {
        set_error("dca_gemm8_mx: too many tiles");
        return DCA_E_BADARG;
    }
    hipLaunchKernelGGL(k_gemm8<true>, dim3((unsigned)blocks), dim3(ETHREADS), lds_bytes, (hipStream_t)stream, p);
    return launch_check("k_gemm8<mx>");
}

int dca_quant_e4m3(const void* x, int dtype, int64_t m, int64_t n, int64_t ld, double scale, void* out, int64_t ldo, void* stream) {
    DCA_ARG(x && out && m >= 0 && n >= 4 && n % 4 == 0 && ld >= n && ld % 4 == 0 && ldo >= n && ldo % 4 == 0 && scale > 0.0);
    DCA_ARG(dtype == DCA_DT_F32 || dtype == DCA_DT_BF16);
    DCA_ARG((uintptr_t)x % (dtype == DCA_DT_F32 ? 16 : 8) == 0 && (uintptr_t)out % 4 == 0);
    if (m == 0) return 0;
    int64_t blocks = (m * (n / 4) + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    if (dtype == DCA_DT_F32)
        hipLaunchKernelGGL(k_quant_e4m3<float>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                           reinterpret_cast<const float*>(x), m, n, ld, (float)scale, reinterpret_cast<uint8_t*>(out), ldo);
    else
        hipLaunchKernelGGL(k_quant_e4m3<uint16_t>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                           reinterpret_cast<const uint16_t*>(x), m, n, ld, (float)scale, reinterpret_cast<uint8_t*>(out), ldo);
    return launch_check("k_quant_e4m3");
}

}  // extern "C"
