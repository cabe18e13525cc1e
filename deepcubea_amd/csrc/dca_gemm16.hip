// dca_gemm16.hip — the dense layers of the cost-to-go network in the NON-parity 16-bit modes (bf16 / fp16 operands, fp32
// accumulation) as ONE hand-written MFMA launch per layer, layer tail included ("persistent fused MLP in
// bf16 ... with fp32 accumulate").
//
// Reference arithmetic (utils/pytorch_models.py:57-86, BatchNorm folded):  v = relu?(x . W^T + b (+ skip)).
// Round 2 ran this mode on the library GEMM (torch._addmm_activation) plus one unfused clamp pass per residual block
// (profiles/r02_nnet_bf16_kernel_stats.csv: 7.7 % of the forward in launch_clamp_scalar, 885 TFLOP/s end to end).  Here
// bias, residual add, ReLU and the rounding to the 16-bit output ride in the epilogue: activations cross HBM once in
// each direction, 2 bytes per element.
//
// Tiling (the f16x3 kernel's, dca_gemm.hip v2, with one operand plane instead of two): workgroup = 256 x 256 outputs, 8
// waves as 2 (M) x 4 (N), each wave 4 x 2 tiles of v_mfma_f32_32x32x16_{bf16,f16} (128 accumulator VGPRs, 32 MFMAs per
// K-step of 64).  The two operand images of a K-step (256 rows x 128 B each = 64 KB) are filled by
// global_load_lds_dwordx4 (LDS-DMA: no staging registers) into one of TWO stages, so the loads of step t+1 fly under the
// MFMAs of step t; one raw s_barrier per K-step.  A DMA instruction writes 64 lanes x 16 B linearly (8 rows of 128 B), so
// the XOR swizzle that makes the ds_read_b128 fragment reads conflict-free — chunk ^ ((row >> 1) & 7) for 128-byte rows —
// is applied to each lane's GLOBAL source address.  Workgroup ids are remapped so that the N tiles sharing an A tile run
// on one XCD.
#include <atomic>
#include <type_traits>

#include "dca_common.h"

namespace dca {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 b16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int QBM = 256, QBN = 256, QBK = 64, QTHREADS = 512;
constexpr int QIMG = 256 * QBK * 2;  // bytes of one operand image (32 KB)
constexpr int QSTAGE = 2 * QIMG;     // A, W
constexpr int QLDS = 2 * QSTAGE;     // two stages: 128 KB

struct Gemm16Args {
    const uint16_t* a;   // [m, lda]
    const uint16_t* w;   // [n, ldw] (row = output unit)
    const float* bias;   // [n] or null
    const uint16_t* skip;  // [m, ldo] or null (same element type)
    uint16_t* out;       // [m, ldo]
    int relu;
    int64_t m;
    int n, k;
    int64_t lda, ldw, ldo;
};

__device__ __forceinline__ uint32_t swz128(uint32_t row, uint32_t chunk) { return row * 128u + ((chunk ^ ((row >> 1) & 7u)) << 4); }

__device__ __forceinline__ uint16_t to_bf16(float f) {  // round to nearest even (what torch does); NaN stays NaN
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40u);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float from_bf16(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
__device__ __forceinline__ uint16_t to_f16(float f) {
    const _Float16 h = (_Float16)f;
    uint16_t r;
    __builtin_memcpy(&r, &h, 2);
    return r;
}
__device__ __forceinline__ float from_f16(uint16_t b) {
    _Float16 h;
    __builtin_memcpy(&h, &b, 2);
    return (float)h;
}

template <bool BF16>
__global__ __launch_bounds__(QTHREADS, 2) void k_gemm16(const Gemm16Args p) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    using frag_t = typename std::conditional<BF16, b16x8, h16x8>::type;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, l31 = lane & 31, h = lane >> 5;
    const int wm = w >> 2, wn = w & 3;
    const int nNt = (p.n + QBN - 1) / QBN;
    const int64_t nMt = (p.m + QBM - 1) / QBM;
    const int64_t bid = blockIdx.x;
    const int64_t slot = bid >> 3;
    const int64_t mt = (slot / nNt) * 8 + (bid & 7);  // the N tiles of one M tile sit on one XCD (workgroup b runs on XCD b % 8)
    const int nt = (int)(slot % nNt);
    if (mt >= nMt) return;
    const int64_t m0 = mt * QBM;
    const int n0 = nt * QBN;

    // LDS-DMA map: instruction q of wave w fills rows [rb*8, rb*8+8) of image q >> 2, rb = (q & 3) * 8 + w; lane i lands
    // on row i >> 3, physical chunk i & 7, and therefore fetches logical chunk (i & 7) ^ ((row >> 1) & 7).  Rows past the
    // matrix edge are clamped to the last row: their products are never stored.
    const uint16_t* src[8];
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int img = q >> 2;
        const uint32_t r = (uint32_t)(((q & 3) * 8 + w) * 8 + (lane >> 3));
        const uint32_t c = (uint32_t)(lane & 7) ^ ((r >> 1) & 7u);
        if (img == 0) {
            int64_t gr = m0 + r;
            gr = gr < p.m ? gr : p.m - 1;
            src[q] = p.a + gr * p.lda + c * 8;
        } else {
            int gn = n0 + (int)r;
            gn = gn < p.n ? gn : p.n - 1;
            src[q] = p.w + (int64_t)gn * p.ldw + c * 8;
        }
    }
    auto issue = [&](int stage, int k0) {
#pragma unroll
        for (int q = 0; q < 8; q++) {
            uint8_t* dst = lds + stage * QSTAGE + (q >> 2) * QIMG + ((q & 3) * 8 + w) * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[q] + k0),
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

    const int nk = p.k / QBK;
    issue(0, 0);
    for (int kt = 0; kt < nk; kt++) {
        // this wave's DMA of step kt has landed and its fragment reads of step kt-1 have returned; the barrier makes that
        // true of every wave — the next issue may overwrite the other stage
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (kt + 1 < nk) issue((kt + 1) & 1, (kt + 1) * QBK);
        const uint8_t* base = lds + (kt & 1) * QSTAGE;
#pragma unroll
        for (int s = 0; s < QBK / 16; s++) {
            const uint32_t c = 2u * s + (uint32_t)h;
            // (typed vector loads, not HIP's uint4 struct: the compiler orders a fragment read behind the LDS-DMA in flight
            // — s_waitcnt vmcnt(0) in front of the first ds_read, no overlap at all — unless type-based alias
            // information tells it the two cannot meet)
            frag_t av[4], wv[2];
#pragma unroll
            for (int i = 0; i < 4; i++) av[i] = *reinterpret_cast<const frag_t*>(base + swz128((uint32_t)(wm * 128 + i * 32 + l31), c));
#pragma unroll
            for (int jn = 0; jn < 2; jn++)
                wv[jn] = *reinterpret_cast<const frag_t*>(base + QIMG + swz128((uint32_t)(wn * 64 + jn * 32 + l31), c));
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int jn = 0; jn < 2; jn++) {
                    if constexpr (BF16)
                        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], wv[jn], acc[i][jn], 0, 0, 0);
                    else
                        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[i], wv[jn], acc[i][jn], 0, 0, 0);
                }
        }
    }

    // epilogue.  Accumulator layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).  Each wave
    // transposes its tile through its own 16 KB of the (now idle) LDS, 32 rows at a time, and leaves with 8-byte accesses:
    // a lane owns 4 consecutive columns of a row (one 8-byte skip load, one 8-byte store; 16 lanes = 128 contiguous bytes).
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // every wave is done with the operand stages
    float* sl = reinterpret_cast<float*>(lds + w * 16384);
    float bv[2];
#pragma unroll
    for (int jn = 0; jn < 2; jn++) {
        const int col = n0 + wn * 64 + jn * 32 + l31;
        bv[jn] = (col < p.n && p.bias) ? p.bias[col] : 0.f;
    }
    const int c4 = (lane & 15) * 4;  // this lane's 4 columns inside the wave's 64
    const int colg = n0 + wn * 64 + c4;
    const bool full4 = colg + 3 < p.n;
    auto cvt_in = [](uint16_t b) { return BF16 ? from_bf16(b) : from_f16(b); };
    auto cvt_out = [](float f) { return BF16 ? to_bf16(f) : to_f16(f); };
#pragma unroll
    for (int i = 0; i < 4; i++) {
#pragma unroll
        for (int jn = 0; jn < 2; jn++)
#pragma unroll
            for (int reg = 0; reg < 16; reg++)
                sl[((reg & 3) + 8 * (reg >> 2) + 4 * h) * 64 + jn * 32 + l31] = acc[i][jn][reg] + bv[jn];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private slice: no barrier, just the wave's own writes
        const int64_t rbase = m0 + wm * 128 + i * 32;
        uint2 sk[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int64_t r = rbase + q * 4 + (lane >> 4);
            sk[q] = make_uint2(0u, 0u);
            if (p.skip && r < p.m && full4) sk[q] = *reinterpret_cast<const uint2*>(p.skip + r * p.ldo + colg);
        }
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int rl = q * 4 + (lane >> 4);
            const int64_t r = rbase + rl;
            const float4 v = *reinterpret_cast<const float4*>(sl + rl * 64 + c4);
            if (r >= p.m) continue;
            float u[4] = {v.x, v.y, v.z, v.w};
            const int64_t o = r * p.ldo + colg;
            if (full4) {
                if (p.skip) {
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
                *reinterpret_cast<uint2*>(p.out + o) = ov;
            } else {  // ragged right edge: element-wise
                for (int e = 0; e < 4 && colg + e < p.n; e++) {
                    float ue = u[e] + (p.skip ? cvt_in(p.skip[o + e]) : 0.f);
                    if (p.relu) ue = fmaxf(ue, 0.f);
                    p.out[o + e] = cvt_out(ue);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the slice is rewritten by the next 32 rows
    }
}

}  // namespace dca

using namespace dca;

extern "C" {

int dca_gemm16(const void* a, int64_t m, int k, int64_t lda, const void* w, int n, int64_t ldw, int dtype, const float* bias,
               const void* skip, int relu, void* out, int64_t ldo, void* stream) {
    DCA_ARG(a && w && out && m >= 0 && n >= 1 && k >= QBK && k % QBK == 0);
    DCA_ARG(dtype == DCA_DT_BF16 || dtype == DCA_DT_F16);
    DCA_ARG(lda >= k && ldw >= k && lda % 8 == 0 && ldw % 8 == 0 && ldo >= n && ldo % 4 == 0);
    DCA_ARG(((uintptr_t)a | (uintptr_t)w) % 16 == 0 && ((uintptr_t)out | (uintptr_t)skip) % 8 == 0);
    if (m == 0) return 0;
    {   // the dynamic-LDS limit is a per-device function attribute: set it once for every device this process uses
        static std::atomic<uint64_t> attr_devs{0};
        int dev = 0;
        DCA_HIP(hipGetDevice(&dev));
        const uint64_t bit = 1ull << (dev & 63);
        if (!(attr_devs.load(std::memory_order_acquire) & bit)) {
            DCA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm16<true>), hipFuncAttributeMaxDynamicSharedMemorySize, QLDS));
            DCA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm16<false>), hipFuncAttributeMaxDynamicSharedMemorySize, QLDS));
            attr_devs.fetch_or(bit, std::memory_order_release);
        }
    }
    Gemm16Args p;
    p.a = reinterpret_cast<const uint16_t*>(a);
    p.w = reinterpret_cast<const uint16_t*>(w);
    p.bias = bias;
    p.skip = reinterpret_cast<const uint16_t*>(skip);
    p.out = reinterpret_cast<uint16_t*>(out);
    p.relu = relu;
    p.m = m;
    p.n = n;
    p.k = k;
    p.lda = lda;
    p.ldw = ldw;
    p.ldo = ldo;
    const int64_t nMt = (m + QBM - 1) / QBM;
    const int64_t nNt = (n + QBN - 1) / QBN;
    const int64_t blocks = ((nMt + 7) / 8) * 8 * nNt;
    if (blocks > 0x7FFFFFFFll) {
        set_error("dca_gemm16: too many tiles");
        return DCA_E_BADARG;
    }
    if (dtype == DCA_DT_BF16)
        hipLaunchKernelGGL(k_gemm16<true>, dim3((unsigned)blocks), dim3(QTHREADS), QLDS, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(k_gemm16<false>, dim3((unsigned)blocks), dim3(QTHREADS), QLDS, (hipStream_t)stream, p);
    return launch_check("k_gemm16");
}

}  // extern "C"
