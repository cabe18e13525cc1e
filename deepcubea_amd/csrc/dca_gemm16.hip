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
// Two schedules over the same tile (dca_gemm16_variant): variant 1 below — two whole K-step stages — and variant 2, the
// default, further down: the ping-pong schedule over half-tile slots (its header has the derivation).
// Tiling (the f16x3 kernel's, dca_gemm.hip, with one operand plane instead of two): workgroup = 256 x 256 outputs, 8
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
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int QBM = 256, QBN = 256, QBK = 64, QTHREADS = 512;
constexpr int QIMG = 256 * QBK * 2;  // bytes of one operand image (32 KB)
constexpr int QSTAGE = 2 * QIMG;     // A, W
constexpr int QLDS = 2 * QSTAGE;     // two stages: 128 KB
constexpr int QLDS_PS = QLDS + 8 * 4096;  // variant 5: + a wave-private 4 KB each for the tail's piece exchange (160 KB)

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
    unsigned long long* prof;  // diagnostics (variant 5): per tile {start, K loop done, tail done} on the 100 MHz wall clock
    int skew_us;               // diagnostics (variant 5): workgroup j of an XCD starts (j % 8) * skew_us / 8 late
};

__device__ __forceinline__ uint32_t swz128(uint32_t row, uint32_t chunk) { return row * 128u + ((chunk ^ ((row >> 1) & 7u)) << 4); }

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
// round to nearest even (what torch does), NaN stays NaN: gfx950's v_cvt_pk_bf16_f32 — one instruction per two values where
// the integer sequence (NaN test, bias add, shift) took six per value: ~700 VALU instructions per wave and tile
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    const bf16x2_t b = __builtin_convertvector(v, bf16x2_t);
    uint32_t u;
    __builtin_memcpy(&u, &b, 4);
    return u;
}
__device__ __forceinline__ uint16_t to_bf16(float f) { return (uint16_t)(pack_bf16x2(f, 0.f) & 0xFFFFu); }
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

// Layer tail, shared by both schedules.  Accumulator layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
template <bool BF16, int NJ>
__device__ __forceinline__ void gemm16_epilogue(const Gemm16Args& p, uint8_t* lds, f32x16 (&acc)[4][NJ], int64_t m0, int n0, int w,
                                                int wm, int wn, int lane, int l31, int h) {
    // Each wave transposes its tile (4 x NJ blocks of 32 x 32) through its own NJ * 8 KB of the (now idle) LDS, 32 rows at a
    // time, and leaves with 8-byte accesses: a lane owns 4 consecutive columns of a row (one 8-byte skip load, one 8-byte
    // store; 16 lanes = 128 contiguous bytes).
    constexpr int CW = NJ * 32, LPR = CW / 4, RPP = 64 / LPR, NP = 32 / RPP;  // columns per wave, lanes per row, rows per pass, passes
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // every wave is done with the operand stages
    float* sl = reinterpret_cast<float*>(lds + w * (CW * 32 * 4));
    float bv[NJ];
#pragma unroll
    for (int jn = 0; jn < NJ; jn++) {
        const int col = n0 + wn * CW + jn * 32 + l31;
        bv[jn] = (col < p.n && p.bias) ? p.bias[col] : 0.f;
    }
    const int c4 = (lane % LPR) * 4;  // this lane's 4 columns inside the wave's CW
    const int colg = n0 + wn * CW + c4;
    const bool full4 = colg + 3 < p.n;
    auto cvt_in = [](uint16_t b) { return BF16 ? from_bf16(b) : from_f16(b); };
    auto cvt_out = [](float f) { return BF16 ? to_bf16(f) : to_f16(f); };
    // (row block as a compile-time constant: left as a loop, the 4 x NJ accumulator blocks would be indexed dynamically —
    // the compiler then keeps all of them in scratch)
    auto rows32 = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
#pragma unroll
        for (int jn = 0; jn < NJ; jn++)
#pragma unroll
            for (int reg = 0; reg < 16; reg++)
                sl[((reg & 3) + 8 * (reg >> 2) + 4 * h) * CW + jn * 32 + l31] = acc[i][jn][reg] + bv[jn];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private slice: no barrier, just the wave's own writes
        const int64_t rbase = m0 + wm * 128 + i * 32;
        uint2 sk[NP];
#pragma unroll
        for (int q = 0; q < NP; q++) {
            const int64_t r = rbase + q * RPP + (lane / LPR);
            sk[q] = make_uint2(0u, 0u);
            if (p.skip && r < p.m && full4) sk[q] = *reinterpret_cast<const uint2*>(p.skip + r * p.ldo + colg);
        }
#pragma unroll
        for (int q = 0; q < NP; q++) {
            const int rl = q * RPP + (lane / LPR);
            const int64_t r = rbase + rl;
            const float4 v = *reinterpret_cast<const float4*>(sl + rl * CW + c4);
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
                if constexpr (BF16) {
                    ov.x = pack_bf16x2(u[0], u[1]);
                    ov.y = pack_bf16x2(u[2], u[3]);
                } else {
                    ov.x = (uint32_t)cvt_out(u[0]) | ((uint32_t)cvt_out(u[1]) << 16);
                    ov.y = (uint32_t)cvt_out(u[2]) | ((uint32_t)cvt_out(u[3]) << 16);
                }
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
    };
    rows32(std::integral_constant<int, 0>{});
    rows32(std::integral_constant<int, 1>{});
    rows32(std::integral_constant<int, 2>{});
    rows32(std::integral_constant<int, 3>{});
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

    gemm16_epilogue<BF16, 2>(p, lds, acc, m0, n0, w, wm, wn, lane, l31, h);
}


// ---------------------------------------------------------------------------------------------------------------------
// Variant 2: the same tile on an 8-phase "ping-pong" schedule (the CDNA4 guide's 256 x 256 template, re-derived for
// 32x32x16 MFMAs and this kernel's DMA map).  What the two-stage loop above leaves on the table: its single
// vmcnt(0) + barrier per K-step drains the LDS-DMA queue (the prefetch distance is one K-step, the tail of the queue is
// always young), and all eight waves read fragments and then issue MFMAs in step, so a SIMD's matrix pipe idles while
// both of its waves wait on LDS.  Here
//   * the two wave rows (waves 0-3 = tile rows 0-127, waves 4-7 = rows 128-255; one wave of each on every SIMD) run
//     ONE BARRIER APART: while one group issues its 8 MFMAs of a phase (s_setprio 1) the other reads its fragments for
//     the next and issues DMA — every SIMD always has a wave in the matrix section;
//   * a K-tile (64 KB) is staged as four 16 KB HALF-TILES, one DMA issue per phase, chosen so that each is read in ONE phase:
//        A01 = rows {0-63, 128-191} (each wave row's first two 32-row blocks), A23 = the other rows,
//        B0  = columns wn*64 + [0,32) of every wave column, B1 = columns wn*64 + [32,64);
//     phase 1 reads B0 + A01 and multiplies them, phase 2 B1 (x A01), phase 3 A23 (x B1), phase 4 reads nothing
//     (A23 x B0, whose fragments stayed in registers);
//   * the DMA queue is never drained and every half-tile gets FIVE phases to land: a phase restages a slot that was last
//     read one or two phases earlier (P1: A23 of tile t+1, P2: A01 of t+2, P3: B0 of t+2, P4: B1 of t+2) and three phases
//     carry a counted wait for the half-tile the NEXT phase reads — s_waitcnt vmcnt(10): all but the five youngest
//     half-tiles have landed.  (First cut, measured: one wait per K-tile, vmcnt(6) in phase 4, B0 re-read in phase 4 — the
//     youngest half-tile of a tile then has three phases to land, and at 1.5 us of DMA latency under load that set the
//     K-tile time: 1.0 PF at K = 5120, no better than the two-stage loop.  Also measured: one phase per 32-deep K-tile
//     over a RING of five 32 KB buffers (all 160 KB of LDS; 12 fragment reads + 16 MFMAs per phase, half the barriers,
//     every tile three whole phases = 1.5 K-steps to land): 1.06 PF at K = 5120, 0.78 at K = 1024 — the coarser phases
//     lose what the longer lead gains; this finer interleave stays.)
// Ordering rules (guide §5, "read a staged buffer one phase AFTER the wait that retires it"):
//   RAW  every wave waits (vmcnt) BEFORE the first barrier of phase p; the half-tile is first read in phase p+1, which
//        either group enters only behind a barrier the other group reached after its own wait.
//   WAR  a wave's fragment reads have RETURNED (lgkmcnt(0)) before the first barrier of the phase that issues them; a slot
//        is restaged at least one phase later, i.e. behind a barrier that both groups' readers of that slot reached after
//        that wait.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int PSLOT = 128 * 128;  // one half-tile: 128 rows x 128 B
constexpr int PBUF = 4 * PSLOT;   // one K-tile: A01 | A23 | B0 | B1
constexpr int PS_A01 = 0, PS_A23 = 1, PS_B0 = 2, PS_B1 = 3;

#define DCA_BAR() asm volatile("s_barrier" ::: "memory")
#define DCA_RD_DONE_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

template <bool BF16>
__global__ __launch_bounds__(QTHREADS, 2) void k_gemm16p(const Gemm16Args p) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    using frag_t = typename std::conditional<BF16, b16x8, h16x8>::type;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, l31 = lane & 31, h = lane >> 5;
    const int wm = w >> 2, wn = w & 3;
    const int nNt = (p.n + QBN - 1) / QBN;
    const int64_t nMt = (p.m + QBM - 1) / QBM;
    const int64_t bid = blockIdx.x;
    const int64_t slot = bid >> 3;
    const int64_t mt = (slot / nNt) * 8 + (bid & 7);
    const int nt = (int)(slot % nNt);
    if (mt >= nMt) return;
    const int64_t m0 = mt * QBM;
    const int n0 = nt * QBN;

    // DMA map: instruction q (0, 1) of wave w fills local rows [(q*8 + w)*8, +8) of a half-tile slot; lane i lands on local
    // row r = that + (i >> 3), physical chunk i & 7, and fetches logical chunk (i & 7) ^ ((r >> 1) & 7) of the matrix row
    // the slot's local row r stands for.
    const uint16_t* src[4][2];
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const uint32_t r = (uint32_t)((q * 8 + w) * 8 + (lane >> 3));
            const uint32_t c = (uint32_t)(lane & 7) ^ ((r >> 1) & 7u);
            if (u < 2) {
                int64_t gr = m0 + (r >> 6) * 128 + (u == PS_A23 ? 64 : 0) + (r & 63);
                gr = gr < p.m ? gr : p.m - 1;
                src[u][q] = p.a + gr * p.lda + c * 8;
            } else {
                int gn = n0 + (int)((r >> 5) * 64 + (u == PS_B1 ? 32 : 0) + (r & 31));
                gn = gn < p.n ? gn : p.n - 1;
                src[u][q] = p.w + (int64_t)gn * p.ldw + c * 8;
            }
        }
    auto issue = [&](int u, int buf, int k0) {
#pragma unroll
        for (int q = 0; q < 2; q++) {
            uint8_t* dst = lds + buf * PBUF + u * PSLOT + (q * 8 + w) * 1024;
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

    // fragment addresses inside a slot: local row = (wave's block) * 32 + l31, logical chunk 2 s + h
    uint32_t foff[4];
#pragma unroll
    for (int s = 0; s < 4; s++) foff[s] = swz128((uint32_t)l31, 2u * s + (uint32_t)h);
    const uint32_t a_row0 = (uint32_t)wm * 64u * 128u;  // A slots: this wave row's 64 local rows
    const uint32_t b_row0 = (uint32_t)wn * 32u * 128u;  // B slots: this wave column's 32 local rows

    frag_t av[2][4], wv0[4], wv1[4];
    auto read_a = [&](const uint8_t* base, int u) {
#pragma unroll
        for (int ii = 0; ii < 2; ii++)
#pragma unroll
            for (int s = 0; s < 4; s++)
                av[ii][s] = *reinterpret_cast<const frag_t*>(base + u * PSLOT + a_row0 + ii * 4096 + foff[s]);
    };
    auto read_b = [&](const uint8_t* base, int u, frag_t (&wv)[4]) {
#pragma unroll
        for (int s = 0; s < 4; s++) wv[s] = *reinterpret_cast<const frag_t*>(base + u * PSLOT + b_row0 + foff[s]);
    };
#define DCA_MMA8(I0, JN, WV)                                                                                          \
    do {                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
        __builtin_amdgcn_s_setprio(1);                                                                                \
        _Pragma("unroll") for (int s = 0; s < 4; s++) _Pragma("unroll") for (int ii = 0; ii < 2; ii++) {              \
            if constexpr (BF16)                                                                                       \
                acc[(I0) + ii][JN] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[ii][s], WV[s], acc[(I0) + ii][JN], 0, 0, 0); \
            else                                                                                                      \
                acc[(I0) + ii][JN] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[ii][s], WV[s], acc[(I0) + ii][JN], 0, 0, 0);  \
        }                                                                                                             \
        __builtin_amdgcn_s_setprio(0);                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
    } while (0)
#define DCA_VMCNT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")

    const int nk = p.k / QBK;
    // one K-tile; N1 / N2: tiles kt+1 / kt+2 exist (compile-time, so the steady-state body is branch-free).  On entry:
    // issued = all of tile kt and A01, B0, B1 of kt+1; landed and visible = A01, B0 of kt.  The vmcnt numbers count the
    // DMA instructions (2 per half-tile) issued AFTER the half-tile being waited for.
    auto tile = [&](int kt, auto n1c, auto n2c) {
        constexpr bool N1 = decltype(n1c)::value, N2 = decltype(n2c)::value;
        const int b = kt & 1;
        const uint8_t* base = lds + b * PBUF;
        // phase 1: (A01, B0); restage A23 of kt+1 (last read: phase 3 of kt-1); retire B1 of kt
        read_b(base, PS_B0, wv0);
        read_a(base, PS_A01);
        if constexpr (N1) {
            issue(PS_A23, b ^ 1, (kt + 1) * QBK);
            DCA_VMCNT(10);  // behind B1(kt): A23(kt), A01 B0 B1 A23 (kt+1)
        } else {
            DCA_VMCNT(2);   // behind B1(kt): A23(kt)
        }
        DCA_RD_DONE_BAR();
        DCA_MMA8(0, 0, wv0);
        DCA_BAR();
        // phase 2: (A01, B1); restage A01 of kt+2 (last read: phase 1); retire A23 of kt
        read_b(base, PS_B1, wv1);
        if constexpr (N2) {
            issue(PS_A01, b, (kt + 2) * QBK);
            DCA_VMCNT(10);  // behind A23(kt): A01 B0 B1 A23 (kt+1), A01(kt+2)
        } else if constexpr (N1) {
            DCA_VMCNT(8);
        } else {
            DCA_VMCNT(0);
        }
        DCA_RD_DONE_BAR();
        DCA_MMA8(0, 1, wv1);
        DCA_BAR();
        // phase 3: (A23, B1); restage B0 of kt+2 (read once, in phase 1: its fragments stay in registers for phase 4)
        read_a(base, PS_A23);
        if constexpr (N2) issue(PS_B0, b, (kt + 2) * QBK);
        DCA_RD_DONE_BAR();
        DCA_MMA8(2, 1, wv1);
        DCA_BAR();
        // phase 4: (A23, B0) from registers; restage B1 of kt+2 (last read: phase 2); retire A01, B0 of kt+1
        if constexpr (N2) {
            issue(PS_B1, b, (kt + 2) * QBK);
            DCA_VMCNT(10);  // behind B0(kt+1): B1 A23 (kt+1), A01 B0 B1 (kt+2)
        } else if constexpr (N1) {
            DCA_VMCNT(4);   // behind B0(kt+1): B1 A23 (kt+1)
        }
        DCA_RD_DONE_BAR();
        DCA_MMA8(2, 0, wv0);
        DCA_BAR();
    };

    issue(PS_A01, 0, 0);
    issue(PS_B0, 0, 0);
    issue(PS_B1, 0, 0);
    issue(PS_A23, 0, 0);
    if (nk > 1) {
        issue(PS_A01, 1, QBK);
        issue(PS_B0, 1, QBK);
        issue(PS_B1, 1, QBK);
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
#undef DCA_VMCNT
#undef DCA_MMA8

    gemm16_epilogue<BF16, 2>(p, lds, acc, m0, n0, w, wm, wn, lane, l31, h);
}

// ---------------------------------------------------------------------------------------------------------------------
// Variant 4: the same 256 x 256 workgroup tile on FOUR waves (2 x 2), each owning 128 x 128 outputs — 4 x 4 blocks of
// v_mfma_f32_32x32x16, 256 accumulator registers (one wave per SIMD: the whole 512-entry register file is the wave's).
// What the shape buys over the 8-wave layouts above: a K-slice of 16 costs a wave 4 + 4 fragment reads for 16 MFMAs (0.5
// ds_read_b128 per MFMA, against 0.75 for 4 x 2 blocks) and nothing is read twice by two waves of one SIMD.  What it
// loses is the partner wave that covers LDS latency, so the wave pipelines itself: two fragment register sets, the reads
// of the next 16-deep slice are in flight while the 16 MFMAs of the current one issue.
// Staging: K-tiles of 32 (64-byte rows; swizzle chunk ^ ((row >> 2) & 3), applied on the global side of the LDS-DMA as
// everywhere in this file), a RING OF FIVE 32 KB slots — all 160 KB of LDS.  Phase h computes K-tile h (two slices) and
// restages the slot phase h-1 read with K-tile h+4: four phases = 4096 MFMA cycles of lead for every DMA.  One barrier
// per phase, placed BETWEEN the two MFMA groups: behind it the wave reads the first slice of tile h+1 while the second
// slice of tile h multiplies.
//   RAW  vmcnt (all but the three youngest tiles have landed: tile h+1 is in) precedes the barrier of phase h; tile h+1 is
//        first read behind it.
//   WAR  lgkmcnt(0) precedes that barrier too: every fragment read of tile h has returned before any wave restages its
//        slot (in phase h+1).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int WTHREADS = 256, WBK = 32;
constexpr int WIMG = 256 * WBK * 2;  // one operand image of a K-tile: 256 rows x 64 B
constexpr int WSLOT = 2 * WIMG;      // A | W
constexpr int WRING = 5;
constexpr int WLDS = WRING * WSLOT;  // 160 KB

template <bool BF16>
__global__ __launch_bounds__(WTHREADS, 1) void k_gemm16w(const Gemm16Args p) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    using frag_t = typename std::conditional<BF16, b16x8, h16x8>::type;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, l31 = lane & 31, h = lane >> 5;
    const int wm = w >> 1, wn = w & 1;
    const int nNt = (p.n + QBN - 1) / QBN;
    const int64_t nMt = (p.m + QBM - 1) / QBM;
    const int64_t bid = blockIdx.x;
    const int64_t slot_id = bid >> 3;
    const int64_t mt = (slot_id / nNt) * 8 + (bid & 7);
    const int nt = (int)(slot_id % nNt);
    if (mt >= nMt) return;
    const int64_t m0 = mt * QBM;
    const int n0 = nt * QBN;

    // DMA map: instruction q (0-7) of wave w fills rows [rb*16, rb*16 + 16) of image q >> 2, rb = (q & 3) * 4 + w; lane i
    // lands on row i >> 2, physical chunk i & 3, and fetches logical chunk (i & 3) ^ ((row >> 2) & 3).
    const uint16_t* src[8];
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const uint32_t r = (uint32_t)(((q & 3) * 4 + w) * 16 + (lane >> 2));
        const uint32_t c = (uint32_t)(lane & 3) ^ ((r >> 2) & 3u);
        if ((q >> 2) == 0) {
            int64_t gr = m0 + r;
            gr = gr < p.m ? gr : p.m - 1;
            src[q] = p.a + gr * p.lda + c * 8;
        } else {
            int gn = n0 + (int)r;
            gn = gn < p.n ? gn : p.n - 1;
            src[q] = p.w + (int64_t)gn * p.ldw + c * 8;
        }
    }
    auto issue = [&](int slot, int k0) {
#pragma unroll
        for (int q = 0; q < 8; q++) {
            uint8_t* dst = lds + slot * WSLOT + (q >> 2) * WIMG + ((q & 3) * 4 + w) * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[q] + k0),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int jn = 0; jn < 4; jn++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][jn][e] = 0.f;

    // fragment addresses inside a slot: row = (wave's block) * 32 + l31, logical chunk 2 s + h  ((row >> 2) & 3 == (l31 >> 2) & 3)
    uint32_t foff[2];
#pragma unroll
    for (int s = 0; s < 2; s++) foff[s] = (uint32_t)l31 * 64u + (((2u * s + (uint32_t)h) ^ (((uint32_t)l31 >> 2) & 3u)) << 4);
    const uint32_t a_row0 = (uint32_t)wm * 128u * 64u;
    const uint32_t b_row0 = (uint32_t)WIMG + (uint32_t)wn * 128u * 64u;

    // The wave has no partner to fill its issue gaps, so the phase is scheduled by hand, one non-MFMA instruction in the
    // shadow of each MFMA (an MFMA holds the matrix pipe for 32 cycles; the wave is free to issue the next fragment read or
    // DMA meanwhile) and every position pinned with sched_barrier:
    //   group A, 16 MFMAs on slice 0:  the 8 fragment reads of slice 1 behind MFMAs 0-7, the 8 DMA instructions of tile
    //            h+4 behind MFMAs 8-15;
    //   group B, 16 MFMAs on slice 1:  MFMAs 0-7, then vmcnt + THE barrier (tile h+1 visible, this slot's reads all
    //            returned), then the 8 reads of slice 0 of tile h+1 behind MFMAs 8-15.
    // MFMAs run in "growing square" order over the 4 x 4 blocks — (0,0) (1,0) (0,1) (1,1) (2,0) (2,1) (0,2) (1,2) (2,2)
    // (3,0) (3,1) (3,2) (0,3) (1,3) (2,3) (3,3) — and the fragments are read in the order that square needs them (a0 b0 a1 b1
    // a2 b2 a3 b3), so MFMA 0 waits for the two oldest reads only.  Fragment reads are inline asm with hand-counted
    // lgkmcnt: for loads it knows about hipcc waits lgkmcnt(0) in front of an MFMA group — the reads just issued for the
    // next slice included.  lgkmcnt(N) before an MFMA = reads issued behind the youngest fragment it needs.
    u32x4 fa0[4], fb0[4], fa1[4], fb1[4];
    const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) uint8_t*)lds);
    const uint32_t fo_a0 = lds0 + a_row0 + foff[0], fo_a1 = lds0 + a_row0 + foff[1];
    const uint32_t fo_b0 = lds0 + b_row0 + foff[0], fo_b1 = lds0 + b_row0 + foff[1];
#define DCA_SB() __builtin_amdgcn_sched_barrier(0)
#define DCA_DSR(D, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(D) : "v"(ADDR), "i"(OFF) : "memory")
    // read k of a slice: 0 a0, 1 b0, 2 a1, 3 b1, 4 a2, 5 b2, 6 a3, 7 b3
#define DCA_RD(K, FA, FB, AA, AB)                                                                                  \
    do {                                                                                                           \
        if constexpr (((K) & 1) == 0)                                                                              \
            DCA_DSR(FA[(K) >> 1], AA, ((K) >> 1) * 2048);                                                          \
        else                                                                                                       \
            DCA_DSR(FB[(K) >> 1], AB, ((K) >> 1) * 2048);                                                          \
        DCA_SB();                                                                                                  \
    } while (0)
#define DCA_WL(N)                                                                                                  \
    do {                                                                                                           \
        asm volatile("s_waitcnt lgkmcnt(" #N ")" ::: "memory");                                                    \
        DCA_SB(); /* (an MFMA has no dependence on the wait: without the fence the scheduler hoists it above) */    \
    } while (0)
#define DCA_MM(FA, FB, I, J)                                                                                       \
    do {                                                                                                           \
        if constexpr (BF16)                                                                                        \
            acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(frag_t, FA[I]),                 \
                                                                __builtin_bit_cast(frag_t, FB[J]), acc[I][J], 0, 0, 0); \
        else                                                                                                       \
            acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(frag_t, FA[I]),                  \
                                                               __builtin_bit_cast(frag_t, FB[J]), acc[I][J], 0, 0, 0); \
        DCA_SB();                                                                                                  \
    } while (0)
#define DCA_VMCNT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
    auto dma1 = [&](int q, int slot, int k0) {
        uint8_t* dst = lds + slot * WSLOT + (q >> 2) * WIMG + ((q & 3) * 4 + w) * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[q] + k0),
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        DCA_SB();
    };

    const int nh = p.k / WBK;
    // one phase.  DMA: tile hh+4 exists (restage the slot phase hh-1 read); NEXT: tile hh+1 exists; VMC: DMA instructions
    // issued behind tile hh+1 (8 per tile, three tiles at most)
    auto phase = [&](int hh, int slot, auto dmac, auto nextc, auto vmc) {
        constexpr bool DMA = decltype(dmac)::value, NEXT = decltype(nextc)::value;
        constexpr int VMC = decltype(vmc)::value;
        const uint32_t sb = (uint32_t)slot * WSLOT;
        const uint32_t aa1 = fo_a1 + sb, ab1 = fo_b1 + sb;
        const int dslot = slot == 0 ? WRING - 1 : slot - 1;
        const int dk0 = (hh + 4) * WBK;
        DCA_SB();
        // ---- group A: slice 0 (its 8 reads were issued behind the last 8 MFMAs of the previous phase)
        DCA_WL(6);
        DCA_MM(fa0, fb0, 0, 0);
        DCA_RD(0, fa1, fb1, aa1, ab1);
        DCA_WL(6);
        DCA_MM(fa0, fb0, 1, 0);
        DCA_RD(1, fa1, fb1, aa1, ab1);
        DCA_WL(6);
        DCA_MM(fa0, fb0, 0, 1);
        DCA_RD(2, fa1, fb1, aa1, ab1);
        DCA_MM(fa0, fb0, 1, 1);
        DCA_RD(3, fa1, fb1, aa1, ab1);
        DCA_WL(7);
        DCA_MM(fa0, fb0, 2, 0);
        DCA_RD(4, fa1, fb1, aa1, ab1);
        DCA_MM(fa0, fb0, 2, 1);
        DCA_RD(5, fa1, fb1, aa1, ab1);
        DCA_WL(8);
        DCA_MM(fa0, fb0, 0, 2);
        DCA_RD(6, fa1, fb1, aa1, ab1);
        DCA_MM(fa0, fb0, 1, 2);
        DCA_RD(7, fa1, fb1, aa1, ab1);
        DCA_MM(fa0, fb0, 2, 2);
        if constexpr (DMA) dma1(0, dslot, dk0);
        DCA_WL(9);
        DCA_MM(fa0, fb0, 3, 0);
        if constexpr (DMA) dma1(1, dslot, dk0);
        DCA_MM(fa0, fb0, 3, 1);
        if constexpr (DMA) dma1(2, dslot, dk0);
        DCA_MM(fa0, fb0, 3, 2);
        if constexpr (DMA) dma1(3, dslot, dk0);
        DCA_WL(8);
        DCA_MM(fa0, fb0, 0, 3);
        if constexpr (DMA) dma1(4, dslot, dk0);
        DCA_MM(fa0, fb0, 1, 3);
        if constexpr (DMA) dma1(5, dslot, dk0);
        DCA_MM(fa0, fb0, 2, 3);
        if constexpr (DMA) dma1(6, dslot, dk0);
        DCA_MM(fa0, fb0, 3, 3);
        if constexpr (DMA) dma1(7, dslot, dk0);
        // ---- group B: slice 1
        DCA_WL(0);  // slice 1 is in: every fragment read of this slot has returned
        DCA_MM(fa1, fb1, 0, 0);
        DCA_MM(fa1, fb1, 1, 0);
        DCA_MM(fa1, fb1, 0, 1);
        DCA_MM(fa1, fb1, 1, 1);
        DCA_MM(fa1, fb1, 2, 0);
        DCA_MM(fa1, fb1, 2, 1);
        DCA_MM(fa1, fb1, 0, 2);
        DCA_MM(fa1, fb1, 1, 2);
        if constexpr (VMC == 24)
            DCA_VMCNT(24);
        else if constexpr (VMC == 16)
            DCA_VMCNT(16);
        else if constexpr (VMC == 8)
            DCA_VMCNT(8);
        else
            DCA_VMCNT(0);
        DCA_BAR();  // tile hh+1 has landed for everyone; nobody reads this slot any more
        DCA_SB();
        const int nslot = slot == WRING - 1 ? 0 : slot + 1;
        const uint32_t sn = (uint32_t)nslot * WSLOT;
        const uint32_t aa0 = fo_a0 + sn, ab0 = fo_b0 + sn;
        DCA_MM(fa1, fb1, 2, 2);
        if constexpr (NEXT) DCA_RD(0, fa0, fb0, aa0, ab0);
        DCA_MM(fa1, fb1, 3, 0);
        if constexpr (NEXT) DCA_RD(1, fa0, fb0, aa0, ab0);
        DCA_MM(fa1, fb1, 3, 1);
        if constexpr (NEXT) DCA_RD(2, fa0, fb0, aa0, ab0);
        DCA_MM(fa1, fb1, 3, 2);
        if constexpr (NEXT) DCA_RD(3, fa0, fb0, aa0, ab0);
        DCA_MM(fa1, fb1, 0, 3);
        if constexpr (NEXT) DCA_RD(4, fa0, fb0, aa0, ab0);
        DCA_MM(fa1, fb1, 1, 3);
        if constexpr (NEXT) DCA_RD(5, fa0, fb0, aa0, ab0);
        DCA_MM(fa1, fb1, 2, 3);
        if constexpr (NEXT) DCA_RD(6, fa0, fb0, aa0, ab0);
        DCA_MM(fa1, fb1, 3, 3);
        if constexpr (NEXT) DCA_RD(7, fa0, fb0, aa0, ab0);
    };

    issue(0, 0);
    if (nh > 1) issue(1, WBK);
    if (nh > 2) issue(2, 2 * WBK);
    if (nh > 3) issue(3, 3 * WBK);
    if (nh > 3)
        DCA_VMCNT(24);
    else if (nh == 3)
        DCA_VMCNT(16);
    else if (nh == 2)
        DCA_VMCNT(8);
    else
        DCA_VMCNT(0);
    DCA_BAR();
    DCA_SB();
    DCA_RD(0, fa0, fb0, fo_a0, fo_b0);
    DCA_RD(1, fa0, fb0, fo_a0, fo_b0);
    DCA_RD(2, fa0, fb0, fo_a0, fo_b0);
    DCA_RD(3, fa0, fb0, fo_a0, fo_b0);
    DCA_RD(4, fa0, fb0, fo_a0, fo_b0);
    DCA_RD(5, fa0, fb0, fo_a0, fo_b0);
    DCA_RD(6, fa0, fb0, fo_a0, fo_b0);
    DCA_RD(7, fa0, fb0, fo_a0, fo_b0);
    {
        using T = std::true_type;
        using F = std::false_type;
        int slot = 0, hh = 0;
        auto adv = [&]() {
            slot = slot == WRING - 1 ? 0 : slot + 1;
            hh++;
        };
        while (hh + 4 < nh) {
            phase(hh, slot, T{}, T{}, std::integral_constant<int, 24>{});
            adv();
        }
        if (hh + 3 < nh) {
            phase(hh, slot, F{}, T{}, std::integral_constant<int, 16>{});
            adv();
        }
        if (hh + 2 < nh) {
            phase(hh, slot, F{}, T{}, std::integral_constant<int, 8>{});
            adv();
        }
        if (hh + 1 < nh) {
            phase(hh, slot, F{}, T{}, std::integral_constant<int, 0>{});
            adv();
        }
        phase(hh, slot, F{}, F{}, std::integral_constant<int, 0>{});
    }
#undef DCA_VMCNT
#undef DCA_MM
#undef DCA_WL
#undef DCA_RD
#undef DCA_DSR
#undef DCA_SB

    gemm16_epilogue<BF16, 4>(p, lds, acc, m0, n0, w, wm, wn, lane, l31, h);
}


// ---------------------------------------------------------------------------------------------------------------------
// Variant 5 (round 5): the ping-pong K loop of variant 2 inside a PERSISTENT workgroup with a register-only layer tail.
// What the profile of variant 2 said (profiles/r04_gemm_timeline.txt): per 256 x 256 tile at K = 1024 the K loop takes
// ~26 us and everything around it ~15 — 3.6 us until the first operands have landed, 10-12 us of tail (accumulators through
// LDS, skip rows in four dependent round trips, stores), 1.6 us until the CU's next workgroup starts.  None of that needs the
// matrix pipe, and none of it overlapped with anything.  Three changes, same products in the same order:
//   * OPERAND ROLES SWAPPED in the MFMA: the weight fragment is the instruction's A operand, the activation fragment its B
//     operand, so D[i][j] has j = lane & 31 = the activation ROW and i (the register index) = the output column.  A lane
//     then owns a piece of ONE output row — and which physical weight row feeds MFMA row i is free to choose (it is only
//     the source address of an LDS-DMA piece): sigma() below makes a lane's 8 consecutive accumulator registers 8
//     CONSECUTIVE output columns (16 bytes of bf16) and the two lane halves adjacent.  The whole tail — bias, skip, ReLU,
//     rounding, store — runs on registers: one 16-byte skip load and one 16-byte store per 8 values, no LDS, no barrier.
//   * PERSISTENT: gridDim = CUs; a workgroup walks tiles slot = j + t * (grid / 8) of its XCD (same XCD mapping as before:
//     the N tiles of one M tile share that XCD's L2).  No per-tile launch gap, no re-derivation of the maps.
//   * CROSS-TILE PREFETCH: the tail does not touch the LDS, so the first seven half-tiles of the NEXT tile are requested
//     before the tail starts and land under it; the next K loop starts with its operands in place.
// vmcnt bookkeeping: loads and stores retire in issue order (one counter, gfx9 family), so the counted waits of the K loop
// stay correct with the tail's loads / stores in the queue: everything older than the N youngest entries has completed, and
// the N youngest are always DMA pieces of the K loop itself (N <= 10 < pieces issued since the tail).  They are merely a
// little stricter than necessary in a tile's first phases (they also wait for the tail's stores to be acknowledged).
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int sigma16(int jn, int i) {
    // MFMA row i = r + 8 q + 4 h (r = reg & 3, q = reg >> 2, h = lane half)  ->  column inside the wave's 64:
    // (jn * 2 + (q >> 1)) * 16 + h * 8 + (q & 1) * 4 + r
    const int r = i & 3, hh = (i >> 2) & 1, q = i >> 3;
    return (jn * 2 + (q >> 1)) * 16 + hh * 8 + (q & 1) * 4 + r;
}

template <bool BF16>
__global__ __launch_bounds__(QTHREADS, 2) void k_gemm16ps(const Gemm16Args p) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    using frag_t = typename std::conditional<BF16, b16x8, h16x8>::type;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, l31 = lane & 31, h = lane >> 5;
    const int wm = w >> 2, wn = w & 3;
    const int nNt = (p.n + QBN - 1) / QBN;
    const int64_t nMt = (p.m + QBM - 1) / QBM;
    const int64_t slots = ((nMt + 7) / 8) * nNt;  // per XCD
    const int xcd = (int)(blockIdx.x & 7);
    const int64_t stride = gridDim.x >> 3;
    const int nk = p.k / QBK;

    const uint16_t* src[4][2];
    auto set_src = [&](int64_t m0, int n0) {
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const uint32_t r = (uint32_t)((q * 8 + w) * 8 + (lane >> 3));
                const uint32_t c = (uint32_t)(lane & 7) ^ ((r >> 1) & 7u);
                if (u < 2) {
                    int64_t gr = m0 + (r >> 6) * 128 + (u == PS_A23 ? 64 : 0) + (r & 63);
                    gr = gr < p.m ? gr : p.m - 1;
                    src[u][q] = p.a + gr * p.lda + c * 8;
                } else {
                    int gn = n0 + (int)(r >> 5) * 64 + sigma16(u == PS_B1 ? 1 : 0, (int)(r & 31));
                    gn = gn < p.n ? gn : p.n - 1;
                    src[u][q] = p.w + (int64_t)gn * p.ldw + c * 8;
                }
            }
    };
    auto issue = [&](int u, int buf, int k0) {
#pragma unroll
        for (int q = 0; q < 2; q++) {
            uint8_t* dst = lds + buf * PBUF + u * PSLOT + (q * 8 + w) * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[u][q] + k0),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };
    auto prologue = [&]() {  // all of K-tile 0 and A01, B0, B1 of K-tile 1
        issue(PS_A01, 0, 0);
        issue(PS_B0, 0, 0);
        issue(PS_B1, 0, 0);
        issue(PS_A23, 0, 0);
        if (nk > 1) {
            issue(PS_A01, 1, QBK);
            issue(PS_B0, 1, QBK);
            issue(PS_B1, 1, QBK);
        }
    };
    auto tile_of = [&](int64_t slot, int64_t& m0, int& n0) -> bool {
        if (slot >= slots) return false;
        const int64_t mt = (slot / nNt) * 8 + xcd;
        m0 = mt * QBM;
        n0 = (int)(slot % nNt) * QBN;
        return true;
    };
    // first tile of this workgroup with rows inside the matrix (slots of an XCD whose M tile lies past the edge are skipped)
    auto next_tile = [&](int64_t& slot, int64_t& m0, int& n0) -> bool {
        for (;; slot += stride) {
            if (!tile_of(slot, m0, n0)) return false;
            if (m0 < p.m) return true;
        }
    };

    uint32_t foff[4];
#pragma unroll
    for (int s = 0; s < 4; s++) foff[s] = swz128((uint32_t)l31, 2u * s + (uint32_t)h);
    const uint32_t a_row0 = (uint32_t)wm * 64u * 128u;
    const uint32_t b_row0 = (uint32_t)wn * 32u * 128u;

    f32x16 acc[4][2];
    frag_t av[2][4], wv0[4], wv1[4];
    auto read_a = [&](const uint8_t* base, int u) {
#pragma unroll
        for (int ii = 0; ii < 2; ii++)
#pragma unroll
            for (int s = 0; s < 4; s++)
                av[ii][s] = *reinterpret_cast<const frag_t*>(base + u * PSLOT + a_row0 + ii * 4096 + foff[s]);
    };
    auto read_b = [&](const uint8_t* base, int u, frag_t (&wv)[4]) {
#pragma unroll
        for (int s = 0; s < 4; s++) wv[s] = *reinterpret_cast<const frag_t*>(base + u * PSLOT + b_row0 + foff[s]);
    };
    // (weight fragment = A operand, activation fragment = B operand: see the header)
#define DCA_MMA8(I0, JN, WV)                                                                                          \
    do {                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
        __builtin_amdgcn_s_setprio(1);                                                                                \
        _Pragma("unroll") for (int s = 0; s < 4; s++) _Pragma("unroll") for (int ii = 0; ii < 2; ii++) {              \
            if constexpr (BF16)                                                                                       \
                acc[(I0) + ii][JN] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WV[s], av[ii][s], acc[(I0) + ii][JN], 0, 0, 0); \
            else                                                                                                      \
                acc[(I0) + ii][JN] = __builtin_amdgcn_mfma_f32_32x32x16_f16(WV[s], av[ii][s], acc[(I0) + ii][JN], 0, 0, 0);  \
        }                                                                                                             \
        __builtin_amdgcn_s_setprio(0);                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
    } while (0)
#define DCA_VMCNT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
    auto tile = [&](int kt, auto n1c, auto n2c) {  // one K-tile: identical to variant 2's
        constexpr bool N1 = decltype(n1c)::value, N2 = decltype(n2c)::value;
        const int b = kt & 1;
        const uint8_t* base = lds + b * PBUF;
        read_b(base, PS_B0, wv0);
        read_a(base, PS_A01);
        if constexpr (N1) {
            issue(PS_A23, b ^ 1, (kt + 1) * QBK);
            DCA_VMCNT(10);
        } else {
            DCA_VMCNT(2);
        }
        DCA_RD_DONE_BAR();
        DCA_MMA8(0, 0, wv0);
        DCA_BAR();
        read_b(base, PS_B1, wv1);
        if constexpr (N2) {
            issue(PS_A01, b, (kt + 2) * QBK);
            DCA_VMCNT(10);
        } else if constexpr (N1) {
            DCA_VMCNT(8);
        } else {
            DCA_VMCNT(0);
        }
        DCA_RD_DONE_BAR();
        DCA_MMA8(0, 1, wv1);
        DCA_BAR();
        read_a(base, PS_A23);
        if constexpr (N2) issue(PS_B0, b, (kt + 2) * QBK);
        DCA_RD_DONE_BAR();
        DCA_MMA8(2, 1, wv1);
        DCA_BAR();
        if constexpr (N2) {
            issue(PS_B1, b, (kt + 2) * QBK);
            DCA_VMCNT(10);
        } else if constexpr (N1) {
            DCA_VMCNT(4);
        }
        DCA_RD_DONE_BAR();
        DCA_MMA8(2, 0, wv0);
        DCA_BAR();
    };

    auto cvt_in = [](uint32_t b) { return BF16 ? from_bf16((uint16_t)b) : from_f16((uint16_t)b); };
    auto cvt_out = [](float f) { return (uint32_t)(BF16 ? to_bf16(f) : to_f16(f)); };
    const bool al16 = ((p.ldo & 7) == 0) && (((uintptr_t)p.out | (uintptr_t)p.skip) & 15) == 0;
    const bool bal16 = ((uintptr_t)p.bias & 15) == 0;

    int64_t slot = blockIdx.x >> 3, m0 = 0;
    int n0 = 0;
    bool have = next_tile(slot, m0, n0);
    if (!have) return;
    if (p.skew_us > 0) {  // (diagnostic: take the CUs' tiles out of lockstep)
        const unsigned long long t0 = wall_clock64();
        const unsigned long long wait = (unsigned long long)((blockIdx.x >> 3) & 7) * (unsigned long long)p.skew_us * 100ull / 8ull;
        while (wall_clock64() - t0 < wait) __builtin_amdgcn_s_sleep(32);
    }
    set_src(m0, n0);
    prologue();
    int64_t tcount = 0;
    while (have) {
        if (p.prof && t == 0) p.prof[(tcount * gridDim.x + blockIdx.x) * 4 + 0] = wall_clock64();
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int jn = 0; jn < 2; jn++)
#pragma unroll
                for (int e = 0; e < 16; e++) acc[i][jn][e] = 0.f;
        // A01, B0 of K-tile 0 have landed (everything older than the 10 youngest pieces: B1 A23 of tile 0, A01 B0 B1 of tile 1)
        if (nk > 1)
            DCA_VMCNT(10);
        else
            DCA_VMCNT(4);
        DCA_BAR();
        if (wm == 1) DCA_BAR();  // the second wave row runs one barrier behind the first
        {
            int kt = 0;
            for (; kt + 2 < nk; kt++) tile(kt, std::true_type{}, std::true_type{});
            if (kt + 1 < nk) {
                tile(kt, std::true_type{}, std::false_type{});
                kt++;
            }
            tile(kt, std::false_type{}, std::false_type{});
        }
        if (wm == 0) DCA_BAR();  // ... and the first waits for it here: nobody reads the operand slots any more
        if (p.prof && t == 0) p.prof[(tcount * gridDim.x + blockIdx.x) * 4 + 1] = wall_clock64();
        const int64_t cm0 = m0;
        const int cn0 = n0;
        // bias of this lane's 4 x 8 columns: requested first, so that it lands under the address work and DMA issue below
        float bv[4][8];
        {
            const int colb = cn0 + wn * 64 + h * 8;
#pragma unroll
            for (int X = 0; X < 4; X++) {
                const int col = colb + X * 16;
                if (p.bias && col + 7 < p.n && bal16) {
                    const float4 b0 = *reinterpret_cast<const float4*>(p.bias + col), b1 = *reinterpret_cast<const float4*>(p.bias + col + 4);
                    bv[X][0] = b0.x, bv[X][1] = b0.y, bv[X][2] = b0.z, bv[X][3] = b0.w;
                    bv[X][4] = b1.x, bv[X][5] = b1.y, bv[X][6] = b1.z, bv[X][7] = b1.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 8; e++) bv[X][e] = (p.bias && col + e < p.n) ? p.bias[col + e] : 0.f;
                }
            }
        }
        slot += stride;
        have = next_tile(slot, m0, n0);
        if (have) {
            set_src(m0, n0);
            prologue();  // lands under the tail below
        }
        // ---- the layer tail.  Lane (l31, h), accumulator block (i, jn), registers 8 x .. 8 x + 7 are 8 consecutive columns
        //      (one 16-byte piece c16 = X * 2 + h, X = jn * 2 + x, of the wave's 128-byte row segment) of row
        //      cm0 + wm * 128 + i * 32 + l31.  A store with one ROW per lane costs a CU 3.5 us per 128 KB tile, one with 8
        //      lanes per 128-byte row segment 1.0 (tools/store_pattern_probe.hip), so the packed pieces of a 32-row block
        //      change hands inside the wave first — through a wave-private 4 KB of the LDS that the operand slots do not
        //      use (no barrier, no conflict with the prefetch above): lane (4 g + x, h) ends up with piece x * 2 + h of rows
        //      4 g .. 4 g + 3.  Skip rows come in by the same route, the other way round.
        uint8_t* tl = lds + 2 * PBUF + w * 4096;
        const int tg = l31 >> 2, tx = l31 & 3;
        auto tl_addr = [&](int row, int c16) { return tl + row * 128 + ((c16 ^ (row & 7)) << 4); };
        const int colw = cn0 + wn * 64;
        auto load_skip = [&](int i, u32x4 (&sk)[4]) {  // coalesced: piece tx * 2 + h of rows 4 tg + j
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int64_t r = cm0 + wm * 128 + i * 32 + 4 * tg + j;
                const int col = colw + (tx * 2 + h) * 8;
                sk[j] = u32x4{0u, 0u, 0u, 0u};
                if (p.skip && r < p.m && col < p.n) {
                    const uint16_t* sp = p.skip + r * p.ldo + col;
                    if (col + 7 < p.n) {
                        if (al16) {
                            sk[j] = *reinterpret_cast<const u32x4*>(sp);
                        } else {
                            const uint2 lo = *reinterpret_cast<const uint2*>(sp), hi = *reinterpret_cast<const uint2*>(sp + 4);
                            sk[j] = u32x4{lo.x, lo.y, hi.x, hi.y};
                        }
                    } else {  // ragged right edge: element-wise
                        uint32_t e16[8];
#pragma unroll
                        for (int e = 0; e < 8; e++) e16[e] = col + e < p.n ? (uint32_t)sp[e] : 0u;
                        sk[j] = u32x4{e16[0] | (e16[1] << 16), e16[2] | (e16[3] << 16), e16[4] | (e16[5] << 16), e16[6] | (e16[7] << 16)};
                    }
                }
            }
        };
        auto rows32 = [&](auto ic, const u32x4 (&sk)[4]) {
            constexpr int i = decltype(ic)::value;
            u32x4 mine[4];  // this lane's own row: skip pieces X * 2 + h
            if (p.skip) {
#pragma unroll
                for (int j = 0; j < 4; j++) *reinterpret_cast<u32x4*>(tl_addr(4 * tg + j, tx * 2 + h)) = sk[j];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int X = 0; X < 4; X++) mine[X] = *reinterpret_cast<const u32x4*>(tl_addr(l31, X * 2 + h));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
#pragma unroll
            for (int X = 0; X < 4; X++) {
                float u[8];
#pragma unroll
                for (int e = 0; e < 8; e++) u[e] = acc[i][X >> 1][8 * (X & 1) + e] + bv[X][e];
                if (p.skip) {
#pragma unroll
                    for (int e = 0; e < 8; e++) u[e] += cvt_in((mine[X][e >> 1] >> (16 * (e & 1))) & 0xFFFFu);
                }
                if (p.relu) {
#pragma unroll
                    for (int e = 0; e < 8; e++) u[e] = fmaxf(u[e], 0.f);
                }
                u32x4 ov;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    if constexpr (BF16)
                        ov[e] = pack_bf16x2(u[2 * e], u[2 * e + 1]);
                    else
                        ov[e] = cvt_out(u[2 * e]) | (cvt_out(u[2 * e + 1]) << 16);
                }
                *reinterpret_cast<u32x4*>(tl_addr(l31, X * 2 + h)) = ov;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            u32x4 q[4];
#pragma unroll
            for (int j = 0; j < 4; j++) q[j] = *reinterpret_cast<const u32x4*>(tl_addr(4 * tg + j, tx * 2 + h));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the slice is rewritten by the next block)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int64_t r = cm0 + wm * 128 + i * 32 + 4 * tg + j;
                const int col = colw + (tx * 2 + h) * 8;
                if (r >= p.m || col >= p.n) continue;
                uint16_t* op = p.out + r * p.ldo + col;
                if (col + 7 < p.n) {
                    if (al16) {
                        *reinterpret_cast<u32x4*>(op) = q[j];
                    } else {
                        *reinterpret_cast<uint2*>(op) = make_uint2(q[j][0], q[j][1]);
                        *reinterpret_cast<uint2*>(op + 4) = make_uint2(q[j][2], q[j][3]);
                    }
                } else {  // ragged right edge: element-wise
                    for (int e = 0; e < 8 && col + e < p.n; e++) op[e] = (uint16_t)((q[j][e >> 1] >> (16 * (e & 1))) & 0xFFFFu);
                }
            }
        };
        {   // skip rows of block i + 1 are in flight while block i is worked on
            u32x4 ska[4], skb[4];
            load_skip(0, ska);
            load_skip(1, skb);
            rows32(std::integral_constant<int, 0>{}, ska);
            load_skip(2, ska);
            rows32(std::integral_constant<int, 1>{}, skb);
            load_skip(3, skb);
            rows32(std::integral_constant<int, 2>{}, ska);
            rows32(std::integral_constant<int, 3>{}, skb);
        }
        if (p.prof && t == 0) {
            p.prof[(tcount * gridDim.x + blockIdx.x) * 4 + 2] = wall_clock64();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            p.prof[(tcount * gridDim.x + blockIdx.x) * 4 + 3] = wall_clock64();
        }
        tcount++;
    }
#undef DCA_VMCNT
#undef DCA_MMA8
}

}  // namespace dca

using namespace dca;

static int g_gemm16_variant = 2;
static unsigned long long* g_gemm16_prof = nullptr;
static int g_gemm16_skew_us = 0;

namespace dca {
// csrc/dca_gemm2.hip: the two-workgroups-per-CU kernels (variant 3 here, variant 4 of dca_f16x3_gemm)
struct Gemm2Args {
    const uint16_t *a, *a2, *w, *w2;
    int64_t m;
    int n, k;
    int64_t lda, ldw, ldo;
    const float* col_scale;
    const float* bias;
    const void* skip;
    float alpha;
    int relu;
    uint16_t *oh, *ol;
    float* x_out;
    int* overflow;
    int skew_ticks;
    int cus;
};
int gemm2_launch(int mode, const Gemm2Args& p, hipStream_t s);
}  // namespace dca

extern "C" {

/* tuning / test hook: 1 = two K-step stages, one drain + barrier per K-step; 2 (default) = the 8-phase ping-pong schedule;
 * 3 = 128 x 256 tiles, two workgroups per CU (dca_gemm2.hip); 4 = four waves x 128 x 128, ring of five K-tiles of 32 */
/* diagnostics of variant 5: knob 1 = device buffer for per-tile wall-clock stamps (4 x u64 per tile and workgroup; 0 = off),
 * knob 2 = start skew in microseconds */
int dca_gemm16_debug(int knob, long long value) {
    if (knob == 1) g_gemm16_prof = reinterpret_cast<unsigned long long*>((uintptr_t)value);
    if (knob == 2) g_gemm16_skew_us = (int)value;
    return 0;
}

int dca_gemm16_variant(int v) {
    DCA_ARG(v >= 1 && v <= 5);
    g_gemm16_variant = v;
    return 0;
}

int dca_gemm16(const void* a, int64_t m, int k, int64_t lda, const void* w, int n, int64_t ldw, int dtype, const float* bias,
               const void* skip, int relu, void* out, int64_t ldo, void* stream) {
    DCA_ARG(a && w && out && m >= 0 && n >= 1 && k >= QBK && k % QBK == 0);
    DCA_ARG(dtype == DCA_DT_BF16 || dtype == DCA_DT_F16);
    DCA_ARG(lda >= k && ldw >= k && lda % 8 == 0 && ldw % 8 == 0 && ldo >= n && ldo % 4 == 0);
    DCA_ARG(((uintptr_t)a | (uintptr_t)w) % 16 == 0 && ((uintptr_t)out | (uintptr_t)skip) % 8 == 0);
    if (m > 0) {  // `out` may alias `skip`, never the operand (its tiles are re-read while other tiles' epilogues write)
        const uintptr_t a0 = (uintptr_t)a, a1 = a0 + ((size_t)(m - 1) * (size_t)lda + (size_t)k) * 2;
        const uintptr_t o0 = (uintptr_t)out, o1 = o0 + ((size_t)(m - 1) * (size_t)ldo + (size_t)n) * 2;
        DCA_ARG(!(o0 < a1 && a0 < o1));
    }
    if (m == 0) return 0;
    if (g_gemm16_variant == 3) {  // 128 x 256 tiles, 4 waves, two workgroups per CU (csrc/dca_gemm2.hip); bit-identical to 1 and 2
        Gemm2Args q;
        q.a = reinterpret_cast<const uint16_t*>(a);
        q.a2 = nullptr;
        q.w = reinterpret_cast<const uint16_t*>(w);
        q.w2 = nullptr;
        q.m = m;
        q.n = n;
        q.k = k;
        q.lda = lda;
        q.ldw = ldw;
        q.ldo = ldo;
        q.col_scale = nullptr;
        q.bias = bias;
        q.skip = skip;
        q.alpha = 1.f;
        q.relu = relu;
        q.oh = reinterpret_cast<uint16_t*>(out);
        q.ol = nullptr;
        q.x_out = nullptr;
        q.overflow = nullptr;
        q.skew_ticks = 0;
        q.cus = 0;
        return gemm2_launch(dtype == DCA_DT_BF16 ? 1 : 2, q, (hipStream_t)stream);
    }
    {   // the dynamic-LDS limit is a per-device function attribute: set it once for every device this process uses
        static std::atomic<uint64_t> attr_devs{0};
        int dev = 0;
        DCA_HIP(hipGetDevice(&dev));
        const uint64_t bit = 1ull << (dev & 63);
        if (!(attr_devs.load(std::memory_order_acquire) & bit)) {
            DCA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm16<true>), hipFuncAttributeMaxDynamicSharedMemorySize, QLDS));
            DCA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm16<false>), hipFuncAttributeMaxDynamicSharedMemorySize, QLDS));
            DCA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm16p<true>), hipFuncAttributeMaxDynamicSharedMemorySize, QLDS));
            DCA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm16p<false>), hipFuncAttributeMaxDynamicSharedMemorySize, QLDS));
            DCA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm16ps<true>), hipFuncAttributeMaxDynamicSharedMemorySize, QLDS_PS));
            DCA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm16ps<false>), hipFuncAttributeMaxDynamicSharedMemorySize, QLDS_PS));
            DCA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm16w<true>), hipFuncAttributeMaxDynamicSharedMemorySize, WLDS));
            DCA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm16w<false>), hipFuncAttributeMaxDynamicSharedMemorySize, WLDS));
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
    p.prof = g_gemm16_prof;
    p.skew_us = g_gemm16_skew_us;
    const int64_t nMt = (m + QBM - 1) / QBM;
    const int64_t nNt = (n + QBN - 1) / QBN;
    const int64_t blocks = ((nMt + 7) / 8) * 8 * nNt;
    if (blocks > 0x7FFFFFFFll) {
        set_error("dca_gemm16: too many tiles");
        return DCA_E_BADARG;
    }
    const dim3 grid((unsigned)blocks), block(QTHREADS);
    if (g_gemm16_variant == 5) {  // persistent: one workgroup per CU (a multiple of 8: the XCD mapping), never more than tiles
        static int cus = 0;
        if (cus == 0) {
            int dev = 0;
            hipDeviceProp_t prop;
            DCA_HIP(hipGetDevice(&dev));
            DCA_HIP(hipGetDeviceProperties(&prop, dev));
            cus = prop.multiProcessorCount > 8 ? (prop.multiProcessorCount / 8) * 8 : 8;
        }
        const unsigned g = (unsigned)(blocks < cus ? blocks : cus);
        if (dtype == DCA_DT_BF16)
            hipLaunchKernelGGL(k_gemm16ps<true>, dim3(g), block, QLDS_PS, (hipStream_t)stream, p);
        else
            hipLaunchKernelGGL(k_gemm16ps<false>, dim3(g), block, QLDS_PS, (hipStream_t)stream, p);
    } else if (g_gemm16_variant == 4) {  // four waves x (128 x 128): K-tiles of 32
        DCA_ARG(k % WBK == 0);
        if (dtype == DCA_DT_BF16)
            hipLaunchKernelGGL(k_gemm16w<true>, grid, dim3(WTHREADS), WLDS, (hipStream_t)stream, p);
        else
            hipLaunchKernelGGL(k_gemm16w<false>, grid, dim3(WTHREADS), WLDS, (hipStream_t)stream, p);
    } else if (g_gemm16_variant == 2) {
        if (dtype == DCA_DT_BF16)
            hipLaunchKernelGGL(k_gemm16p<true>, grid, block, QLDS, (hipStream_t)stream, p);
        else
            hipLaunchKernelGGL(k_gemm16p<false>, grid, block, QLDS, (hipStream_t)stream, p);
    } else {
        if (dtype == DCA_DT_BF16)
            hipLaunchKernelGGL(k_gemm16<true>, grid, block, QLDS, (hipStream_t)stream, p);
        else
            hipLaunchKernelGGL(k_gemm16<false>, grid, block, QLDS, (hipStream_t)stream, p);
    }
    return launch_check("k_gemm16");
}

}  // extern "C"
