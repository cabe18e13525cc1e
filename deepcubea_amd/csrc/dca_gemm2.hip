// dca_gemm2.hip — the dense layers of the cost-to-go network on the "two workgroups per CU" structure (round 4).
//
// What the 256 x 256 / 8-wave kernels of dca_gemm.hip / dca_gemm16.hip leave on the table (profiles/r03_gemm_bench.txt,
// DESIGN §4.5): ONE workgroup owns a CU, its eight waves move in step through barriers, so while a tile's tail runs —
// accumulators through LDS, residual rows in, results out: 28 of 113 us per f16x3 tile at K = 1024, 15 of 41 for bf16 —
// the matrix pipes of that CU are idle, and a layer whose HBM time (output + residual traffic) is as long as its MFMA
// time (the 1024-wide layers) takes the SUM of the two instead of the maximum.
//
// Here a workgroup is 4 waves (256 threads, <= 256 VGPRs each) on a 128 x 256 output tile, 72 KB of LDS — so TWO
// workgroups share a CU (2 waves per SIMD, one of each): one workgroup's tail (and its barriers, and the latency of its
// fragment reads) runs under the other's MFMAs.  The workgroups of a CU are started half a tile apart (the second wave of
// the dispatch sleeps for half a K loop once) so that their tails do not coincide.
//   * wave tile 64 x 128: 2 x 4 blocks of v_mfma_f32_32x32x16 (128 accumulator VGPRs); 6 fragment reads per 8 block
//     products — the ratio of the 128 x 64 wave tile of the 8-wave kernels;
//   * K is walked in stages of ONE 64-byte row per operand row — 32 bf16 / fp16 elements, or 16 elements of BOTH fp16
//     planes of the f16x3 mode side by side (high | low) — 24 KB per stage (A 128 rows, W 256 rows), a RING of three
//     stages filled by global_load_lds_dwordx4 (LDS-DMA) two stages ahead: one counted s_waitcnt vmcnt(6) + one raw
//     s_barrier per stage, the DMA queue never drained inside the loop;
//   * 64-byte LDS rows, XOR swizzle chunk ^ ((row >> 2) & 3) applied to each lane's GLOBAL source address (the DMA
//     destination is lane-linear), conflict-free ds_read_b128 fragment reads (the geometry of dca_gemm.hip's v2);
//   * the layer tail in the epilogue, through a per-wave LDS transposition (the operand ring is idle by then): every
//     global access 8 or 16 bytes wide, 256 or 512 contiguous bytes per row and wave.
// Results: the products of an accumulator are issued in the order of the 8-wave kernels (K ascending; f16x3: low x high,
// high x low, high x high per 16-deep step) and the tails do the same arithmetic — outputs are BIT-IDENTICAL to variants
// 2 / 3 of dca_f16x3_gemm and 1 / 2 of dca_gemm16, which is the race screen (tests/test_gemm_hip.py).
// Reference arithmetic: utils/pytorch_models.py:57-86 (BatchNorm folded): v = relu?(x . W^T + b (+ skip)).
#include <atomic>
#include <type_traits>

#include "dca_common.h"

namespace dca {

typedef _Float16 xh16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 xb16x8 __attribute__((ext_vector_type(8)));
typedef float xf32x16 __attribute__((ext_vector_type(16)));

constexpr int XBM = 128, XBN = 256, XTHREADS = 256;
constexpr int XSTAGE = (XBM + XBN) * 64;   // bytes of one stage: A image (128 rows x 64 B) then W image (256 rows x 64 B)
constexpr int XA_IMG = XBM * 64;
constexpr int XNSTAGE = 3;
constexpr int XLDS = XNSTAGE * XSTAGE;     // 72 KB: two workgroups per CU

enum { XM_F16X3 = 0, XM_BF16 = 1, XM_F16 = 2 };

struct Gemm2Args {
    // operands: XM_F16X3: a / a2 = high / low fp16 planes [m, lda], w / w2 = high / low weight planes [n, ldw];
    //           XM_BF16 / XM_F16: a [m, lda], w [n, ldw]; a2 = w2 = null
    const uint16_t *a, *a2, *w, *w2;
    int64_t m;
    int n, k;
    int64_t lda, ldw, ldo;
    // tail
    const float* col_scale;  // f16x3: [n] or null
    const float* bias;       // [n] or null
    const void* skip;        // f16x3: fp32 [m, ldo]; 16-bit modes: same type as the operands [m, ldo]; or null
    float alpha;             // f16x3
    int relu;
    uint16_t *oh, *ol;       // f16x3: result planes [m, ldo] or null; 16-bit modes: oh = result [m, ldo]
    float* x_out;            // f16x3: fp32 result or null
    int* overflow;           // f16x3
    int skew_ticks;          // wall-clock ticks (100 MHz) the second dispatch wave of workgroups waits before it starts
    int cus;                 // CUs of the device: workgroups [cus, 2 cus) are every CU's second resident workgroup
};

__device__ __forceinline__ uint32_t xswz(uint32_t row, uint32_t chunk) { return row * 64u + ((chunk ^ ((row >> 2) & 3u)) << 4); }

__device__ __forceinline__ uint16_t x_to_bf16(float f) {  // round to nearest even; NaN stays NaN
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40u);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float x_from_bf16(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
__device__ __forceinline__ uint16_t x_to_f16(float f) {
    const _Float16 h = (_Float16)f;
    uint16_t r;
    __builtin_memcpy(&r, &h, 2);
    return r;
}
__device__ __forceinline__ float x_from_f16(uint16_t b) {
    _Float16 h;
    __builtin_memcpy(&h, &b, 2);
    return (float)h;
}

#define XBAR_WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)\n\ts_barrier" ::: "memory")

template <int MODE>
__global__ __launch_bounds__(XTHREADS, 2) void k_gemm2(const Gemm2Args p) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, l31 = lane & 31, h = lane >> 5;
    const int wm = w >> 1, wn = w & 1;
    const int nNt = (p.n + XBN - 1) / XBN;
    const int64_t nMt = (p.m + XBM - 1) / XBM;
    const int64_t bid = blockIdx.x;
    const int64_t slot = bid >> 3;
    const int64_t mt = (slot / nNt) * 8 + (bid & 7);  // the N tiles of one M tile sit on one XCD (workgroup b runs on XCD b % 8)
    const int nt = (int)(slot % nNt);
    if (mt >= nMt) return;
    const int64_t m0 = mt * XBM;
    const int n0 = nt * XBN;

    // Start-up skew: the dispatcher fills every CU with its first workgroup, then with its second — both would walk their
    // K loops and reach their tails together, for the whole launch (equal tiles take equal time).  The second one waits
    // half a K loop once; from then on one workgroup's tail runs under the other's MFMAs.
    if (p.skew_ticks > 0 && bid >= p.cus && bid < 2 * p.cus) {
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < (unsigned long long)p.skew_ticks) __builtin_amdgcn_s_sleep(32);
    }

    // LDS-DMA map.  Instruction q of wave w fills 16 rows of an image (16 x 64 B = 1 KB, lane-linear): q = 0, 1 -> A rows
    // (q * 4 + w) * 16 ...; q = 2 .. 5 -> W rows ((q - 2) * 4 + w) * 16 ....  Lane i lands on row i >> 2, physical chunk
    // i & 3, and therefore fetches logical chunk c = (i & 3) ^ ((row >> 2) & 3) of its matrix row:
    //   16-bit modes: elements [8 c, 8 c + 8) of the stage's 32;
    //   f16x3: c = 0, 1 -> the high plane's elements [8 c, +8) of the stage's 16; c = 2, 3 -> the low plane's [8 (c - 2), +8).
    // Rows past the matrix edge are clamped to the last row: their products are never stored.
    constexpr int KSTEP = MODE == XM_F16X3 ? 16 : 32;  // elements of K per stage
    const uint16_t* src[6];
#pragma unroll
    for (int q = 0; q < 6; q++) {
        const bool isa = q < 2;
        const uint32_t r = (uint32_t)(((isa ? q : q - 2) * 4 + w) * 16 + (lane >> 2));
        const uint32_t c = (uint32_t)(lane & 3) ^ ((r >> 2) & 3u);
        int64_t grow;
        if (isa) {
            grow = m0 + r;
            grow = grow < p.m ? grow : p.m - 1;
        } else {
            grow = n0 + (int64_t)r;
            grow = grow < p.n ? grow : p.n - 1;
        }
        const int64_t ld = isa ? p.lda : p.ldw;
        if constexpr (MODE == XM_F16X3) {
            const uint16_t* plane = isa ? (c < 2 ? p.a : p.a2) : (c < 2 ? p.w : p.w2);
            src[q] = plane + grow * ld + (c & 1u) * 8;
        } else {
            src[q] = (isa ? p.a : p.w) + grow * ld + c * 8;
        }
    }
    auto issue = [&](int stage, int kt) {
        uint8_t* sb = lds + stage * XSTAGE;
#pragma unroll
        for (int q = 0; q < 6; q++) {
            uint8_t* dst = sb + (q < 2 ? (q * 4 + w) * 1024 : XA_IMG + ((q - 2) * 4 + w) * 1024);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[q] + (int64_t)kt * KSTEP),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };

    xf32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int jn = 0; jn < 4; jn++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][jn][e] = 0.f;

    // fragment offsets inside a stage: A rows wm * 64 + i * 32 + l31, W rows wn * 128 + jn * 32 + l31; logical chunks
    // 16-bit modes: 2 s + h (s = 0, 1: the two 16-deep steps of the stage); f16x3: h (high plane), 2 + h (low plane).
    // (row >> 2) & 3 only depends on l31 here (the block offsets are multiples of 32), so the four chunk offsets are shared.
    // Both layouts read the same two chunks per lane: h and 2 + h (16-bit modes: the stage's two 16-deep steps; f16x3: the
    // high and the low plane of its one step).
    const uint32_t off0 = xswz((uint32_t)l31, (uint32_t)h), off1 = xswz((uint32_t)l31, 2u + (uint32_t)h);
    const uint32_t a_base = (uint32_t)(wm * 64) * 64u;
    const uint32_t b_base = (uint32_t)XA_IMG + (uint32_t)(wn * 128) * 64u;

    auto compute = [&](const uint8_t* sb) {
        if constexpr (MODE == XM_F16X3) {
            xh16x8 ah[2], al[2], wh[4], wl[4];
#pragma unroll
            for (int i = 0; i < 2; i++) {
                ah[i] = *reinterpret_cast<const xh16x8*>(sb + a_base + i * 2048 + off0);
                al[i] = *reinterpret_cast<const xh16x8*>(sb + a_base + i * 2048 + off1);
            }
#pragma unroll
            for (int jn = 0; jn < 4; jn++) {
                wh[jn] = *reinterpret_cast<const xh16x8*>(sb + b_base + jn * 2048 + off0);
                wl[jn] = *reinterpret_cast<const xh16x8*>(sb + b_base + jn * 2048 + off1);
            }
            __builtin_amdgcn_s_setprio(1);
            // per accumulator: low x high, high x low, high x high (the order of dca_f16x3_gemm's other variants); the
            // eight accumulators interleaved so that no MFMA waits on the one before it
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int jn = 0; jn < 4; jn++) acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], wh[jn], acc[i][jn], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int jn = 0; jn < 4; jn++) acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], wl[jn], acc[i][jn], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int jn = 0; jn < 4; jn++) acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], wh[jn], acc[i][jn], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        } else {
            using frag_t = typename std::conditional<MODE == XM_BF16, xb16x8, xh16x8>::type;
            frag_t av[2][2], wv[2][4];
#pragma unroll
            for (int s = 0; s < 2; s++) {
#pragma unroll
                for (int i = 0; i < 2; i++) av[s][i] = *reinterpret_cast<const frag_t*>(sb + a_base + i * 2048 + (s ? off1 : off0));
#pragma unroll
                for (int jn = 0; jn < 4; jn++) wv[s][jn] = *reinterpret_cast<const frag_t*>(sb + b_base + jn * 2048 + (s ? off1 : off0));
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s = 0; s < 2; s++)
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int jn = 0; jn < 4; jn++) {
                        if constexpr (MODE == XM_BF16)
                            acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[s][i], wv[s][jn], acc[i][jn], 0, 0, 0);
                        else
                            acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[s][i], wv[s][jn], acc[i][jn], 0, 0, 0);
                    }
            __builtin_amdgcn_s_setprio(0);
        }
    };

    // K loop.  Before the barrier of stage kt a wave has waited for ITS DMA of that stage (all but the 6 instructions of
    // stage kt + 1 retired) and for its fragment reads of stage kt - 1 (lgkmcnt); behind the barrier that is true of every
    // wave, so stage kt may be read and the slot of stage kt - 1 may be refilled with stage kt + 2.
    const int nk = p.k / KSTEP;
    issue(0, 0);
    if (nk > 1) issue(1, 1);
    int st = 0;
    for (int kt = 0; kt + 1 < nk; kt++) {
        XBAR_WAIT(6);
        if (kt + 2 < nk) issue(st == 0 ? 2 : st - 1, kt + 2);
        compute(lds + st * XSTAGE);
        st = st == 2 ? 0 : st + 1;
    }
    XBAR_WAIT(0);
    compute(lds + st * XSTAGE);

    // ---- layer tail.  Accumulator layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).  Each wave
    // transposes 32 rows x 128 columns at a time through its own 16 KB of the (now idle) ring and leaves with wide accesses:
    // a lane owns 4 consecutive columns of a row, 32 lanes cover the wave's 128 columns, two rows per wave instruction.
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // every wave is done with the operand stages
    float* sl = reinterpret_cast<float*>(lds + w * 16384);
    float cs[4], bv[4];
#pragma unroll
    for (int jn = 0; jn < 4; jn++) {
        const int col = n0 + wn * 128 + jn * 32 + l31;
        const bool cv = col < p.n;
        if constexpr (MODE == XM_F16X3)
            cs[jn] = cv ? (p.col_scale ? p.alpha * p.col_scale[col] : p.alpha) : 0.f;
        else
            cs[jn] = 1.f;
        bv[jn] = (cv && p.bias) ? p.bias[col] : 0.f;
    }
    const int c4 = (lane & 31) * 4;  // this lane's 4 columns inside the wave's 128
    const int colg = n0 + wn * 128 + c4;
    const bool full4 = colg + 3 < p.n;
    bool ovf = false;
#pragma unroll
    for (int i = 0; i < 2; i++) {
#pragma unroll
        for (int jn = 0; jn < 4; jn++)
#pragma unroll
            for (int reg = 0; reg < 16; reg++) {
                const float v = MODE == XM_F16X3 ? acc[i][jn][reg] * cs[jn] + bv[jn] : acc[i][jn][reg] + bv[jn];
                sl[((reg & 3) + 8 * (reg >> 2) + 4 * h) * 128 + jn * 32 + l31] = v;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private slice: no barrier, just the wave's own writes
        const int64_t rbase = m0 + wm * 64 + i * 32;
        if constexpr (MODE == XM_F16X3) {
            typedef _Float16 h4 __attribute__((ext_vector_type(4)));
            const float* skip = reinterpret_cast<const float*>(p.skip);
#pragma unroll
            for (int half = 0; half < 2; half++) {  // 8 rows x 2 per half: the skip loads of a half are in flight together
                float4 sk[8];
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const int64_t r = rbase + (half * 8 + q) * 2 + h;
                    sk[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (skip && r < p.m && full4) sk[q] = *reinterpret_cast<const float4*>(skip + r * p.ldo + colg);
                }
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const int rl = (half * 8 + q) * 2 + h;
                    const int64_t r = rbase + rl;
                    const float4 v = *reinterpret_cast<const float4*>(sl + rl * 128 + c4);
                    if (r >= p.m) continue;
                    float u[4] = {v.x + sk[q].x, v.y + sk[q].y, v.z + sk[q].z, v.w + sk[q].w};
                    const int64_t o = r * p.ldo + colg;
                    if (full4) {
                        h4 hi, lo;
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            if (p.relu) u[e] = fmaxf(u[e], 0.f);
                            ovf |= !(fabsf(u[e]) <= 60000.0f);
                            hi[e] = (_Float16)u[e];
                            lo[e] = (_Float16)(u[e] - (float)hi[e]);
                        }
                        if (p.x_out) *reinterpret_cast<float4*>(p.x_out + o) = make_float4(u[0], u[1], u[2], u[3]);
                        if (p.oh) {
                            *reinterpret_cast<h4*>(p.oh + o) = hi;
                            *reinterpret_cast<h4*>(p.ol + o) = lo;
                        }
                    } else {  // ragged right edge: element-wise
                        for (int e = 0; e < 4 && colg + e < p.n; e++) {
                            float ue = u[e] + (skip ? skip[o + e] : 0.f);
                            if (p.relu) ue = fmaxf(ue, 0.f);
                            ovf |= !(fabsf(ue) <= 60000.0f);
                            if (p.x_out) p.x_out[o + e] = ue;
                            if (p.oh) {
                                const _Float16 hh = (_Float16)ue;
                                p.oh[o + e] = x_to_f16((float)hh);
                                p.ol[o + e] = x_to_f16(ue - (float)hh);
                            }
                        }
                    }
                }
            }
        } else {
            const uint16_t* skip = reinterpret_cast<const uint16_t*>(p.skip);
            auto cvt_in = [](uint16_t b) { return MODE == XM_BF16 ? x_from_bf16(b) : x_from_f16(b); };
            auto cvt_out = [](float f) { return MODE == XM_BF16 ? x_to_bf16(f) : x_to_f16(f); };
            uint2 sk[16];
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const int64_t r = rbase + q * 2 + h;
                sk[q] = make_uint2(0u, 0u);
                if (skip && r < p.m && full4) sk[q] = *reinterpret_cast<const uint2*>(skip + r * p.ldo + colg);
            }
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const int rl = q * 2 + h;
                const int64_t r = rbase + rl;
                const float4 v = *reinterpret_cast<const float4*>(sl + rl * 128 + c4);
                if (r >= p.m) continue;
                float u[4] = {v.x, v.y, v.z, v.w};
                const int64_t o = r * p.ldo + colg;
                if (full4) {
                    if (skip) {
                        u[0] += cvt_in((uint16_t)(sk[q].x & 0xFFFFu));
                        u[1] += cvt_in((uint16_t)(sk[q].x >> 16));
                        u[2] += cvt_in((uint16_t)(sk[q].y & 0xFFFFu));
                        u[3] += cvt_in((uint16_t)(sk[q].y >> 16));
                    }
                    if (p.relu) {
#pragma unroll
                        for (int e = 0; e < 4; e++) u[e] = fmaxf(u[e], 0.f);
                    }
                    uint2 ov;
                    ov.x = (uint32_t)cvt_out(u[0]) | ((uint32_t)cvt_out(u[1]) << 16);
                    ov.y = (uint32_t)cvt_out(u[2]) | ((uint32_t)cvt_out(u[3]) << 16);
                    *reinterpret_cast<uint2*>(p.oh + o) = ov;
                } else {  // ragged right edge: element-wise
                    for (int e = 0; e < 4 && colg + e < p.n; e++) {
                        float ue = u[e] + (skip ? cvt_in(skip[o + e]) : 0.f);
                        if (p.relu) ue = fmaxf(ue, 0.f);
                        p.oh[o + e] = cvt_out(ue);
                    }
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the slice is rewritten by the next 32 rows
    }
    if constexpr (MODE == XM_F16X3) {
        if (ovf && p.oh && p.overflow) *p.overflow = 1;
    }
}
#undef XBAR_WAIT

}  // namespace dca

using namespace dca;

// start-up skew of the second dispatch wave, in 1/16ths of the estimated K-loop time (0 = none; dca_gemm2_skew)
static int g_gemm2_skew16 = 8;

namespace dca {

int gemm2_launch(int mode, const Gemm2Args& p0, hipStream_t s) {
    {   // the dynamic-LDS limit is a per-device function attribute: set it once for every device this process uses
        static std::atomic<uint64_t> attr_devs{0};
        int dev = 0;
        DCA_HIP(hipGetDevice(&dev));
        const uint64_t bit = 1ull << (dev & 63);
        if (!(attr_devs.load(std::memory_order_acquire) & bit)) {
            DCA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm2<XM_F16X3>), hipFuncAttributeMaxDynamicSharedMemorySize, XLDS));
            DCA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm2<XM_BF16>), hipFuncAttributeMaxDynamicSharedMemorySize, XLDS));
            DCA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm2<XM_F16>), hipFuncAttributeMaxDynamicSharedMemorySize, XLDS));
            attr_devs.fetch_or(bit, std::memory_order_release);
        }
    }
    Gemm2Args p = p0;
    const int64_t nMt = (p.m + XBM - 1) / XBM;
    const int64_t nNt = (p.n + XBN - 1) / XBN;
    const int64_t blocks = ((nMt + 7) / 8) * 8 * nNt;
    if (blocks > 0x7FFFFFFFll) {
        set_error("dca_gemm2: too many tiles");
        return DCA_E_BADARG;
    }
    // K-loop time of one tile when it has the CU's matrix pipes to itself: 128 x 256 x k MACs (x 3 products for f16x3) at
    // ~2048 MACs per clock and CU, ~2 GHz -> ticks of the 100 MHz wall clock; the skew is a fraction of it, and only worth
    // anything when a CU sees several tiles per slot
    const double macs = 128.0 * 256.0 * (double)p.k * (mode == XM_F16X3 ? 3.0 : 1.0);
    const double loop_ticks = macs / 2048.0 / 2.0e9 * 1.0e8;
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    p.cus = cus;
    p.skew_ticks = blocks > 4 * (int64_t)cus ? (int)(loop_ticks * (double)g_gemm2_skew16 / 16.0) : 0;
    if (mode == XM_F16X3)
        hipLaunchKernelGGL(k_gemm2<XM_F16X3>, dim3((unsigned)blocks), dim3(XTHREADS), XLDS, s, p);
    else if (mode == XM_BF16)
        hipLaunchKernelGGL(k_gemm2<XM_BF16>, dim3((unsigned)blocks), dim3(XTHREADS), XLDS, s, p);
    else
        hipLaunchKernelGGL(k_gemm2<XM_F16>, dim3((unsigned)blocks), dim3(XTHREADS), XLDS, s, p);
    return launch_check("k_gemm2");
}

}  // namespace dca

extern "C" {

/* tuning hook: start-up skew of every CU's second workgroup in 1/16ths of a tile's K-loop time (default 8 = half; 0 = none) */
int dca_gemm2_skew(int sixteenths) {
    DCA_ARG(sixteenths >= 0 && sixteenths <= 64);
    g_gemm2_skew16 = sixteenths;
    return 0;
}

}  // extern "C"
