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
// Three kernels over the same tile (dca_gemm16_variant): variant 1 below — two whole K-step stages, the plain reference —,
// variant 2 further down: the ping-pong schedule over half-tile slots (its header has the derivation) with the general tail,
// and variant 3, the default: that schedule with swapped operand roles and a lean tail for the network's own layer forms.
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
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// The matrix instruction of all three kernels is v_mfma_f32_16x16x32_{bf16,f16} (round 6; it was 32x32x16 until then).  Both
// shapes peak at the same rate, but on RANDOM operands the chip runs at its power limit and the 32x32x16 form draws more per
// flop: MFMAs alone, from registers, 8 waves per CU, sustain 1780 TFLOP/s with 32x32x16 and 2030 with 16x16x32 (2466 for both on
// all-zero operands; tools/mfma_power_probe.hip, profiles/r06_mfma_power_probe.txt) — hipBLASLt's gfx950 kernels are built on
// 16x16 too ("MI16x16x1" in their names), which is why they lost 7 % to random data where these kernels lost 26 %.
// Fragment of a 16-row block for the K step t (32 deep) of a 64-deep K-tile: lane (g = lane >> 4, j = lane & 15) holds row j,
// 16-byte chunk 4 t + g of the 128-byte row — the same LDS image, DMA map and swizzle as before (the four lane groups of a
// ds_read_b128 still meet 16 different (row parity, slot) pairs: conflict-free).  D = A . B: lane (g, j) holds
// D[4 g + r][j], r = 0..3.
template <bool BF16, typename F>
__device__ __forceinline__ f32x4 mma16(const F& a, const F& b, const f32x4& c) {
    if constexpr (BF16)
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

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

// Layer tail, shared by both schedules.  Accumulator layout (activations = A operand): acc[ib][jb][r] = row ib * 16 + 4 g + r,
// column jb * 16 + j of the wave's 128 x 64 (g = lane >> 4, j = lane & 15).  NJ counts the wave's 32-column groups.
template <bool BF16, int NJ>
__device__ __forceinline__ void gemm16_epilogue(const Gemm16Args& p, uint8_t* lds, f32x4 (&acc)[8][2 * NJ], int64_t m0, int n0, int w,
                                                int wm, int wn, int lane, int l31, int h) {
    const int g4 = lane >> 4, j16 = lane & 15;
    // Each wave transposes its tile (4 x NJ blocks of 32 x 32) through its own NJ * 8 KB of the (now idle) LDS, 32 rows at a
    // time, and leaves with 8-byte accesses: a lane owns 4 consecutive columns of a row (one 8-byte skip load, one 8-byte
    // store; 16 lanes = 128 contiguous bytes).
    constexpr int CW = NJ * 32, LPR = CW / 4, RPP = 64 / LPR, NP = 32 / RPP;  // columns per wave, lanes per row, rows per pass, passes
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // every wave is done with the operand stages
    float* sl = reinterpret_cast<float*>(lds + w * (CW * 32 * 4));
    float bv[2 * NJ];
#pragma unroll
    for (int jb = 0; jb < 2 * NJ; jb++) {
        const int col = n0 + wn * CW + jb * 16 + j16;
        bv[jb] = (col < p.n && p.bias) ? p.bias[col] : 0.f;
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
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int jb = 0; jb < 2 * NJ; jb++)
#pragma unroll
                for (int r = 0; r < 4; r++) sl[(16 * b + 4 * g4 + r) * CW + jb * 16 + j16] = acc[2 * i + b][jb][r] + bv[jb];
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

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int jn = 0; jn < 4; jn++)
#pragma unroll
            for (int e = 0; e < 4; e++) acc[i][jn][e] = 0.f;
    const int g4 = lane >> 4, j16 = lane & 15;

    const int nk = p.k / QBK;
    issue(0, 0);
    for (int kt = 0; kt < nk; kt++) {
        // this wave's DMA of step kt has landed and its fragment reads of step kt-1 have returned; the barrier makes that
        // true of every wave — the next issue may overwrite the other stage
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (kt + 1 < nk) issue((kt + 1) & 1, (kt + 1) * QBK);
        const uint8_t* base = lds + (kt & 1) * QSTAGE;
#pragma unroll
        for (int s = 0; s < QBK / 32; s++) {
            const uint32_t c = 4u * s + (uint32_t)g4;
            // (typed vector loads, not HIP's uint4 struct: the compiler orders a fragment read behind the LDS-DMA in flight
            // — s_waitcnt vmcnt(0) in front of the first ds_read, no overlap at all — unless type-based alias
            // information tells it the two cannot meet)
            frag_t av[8], wv[4];
#pragma unroll
            for (int i = 0; i < 8; i++) av[i] = *reinterpret_cast<const frag_t*>(base + swz128((uint32_t)(wm * 128 + i * 16 + j16), c));
#pragma unroll
            for (int jn = 0; jn < 4; jn++)
                wv[jn] = *reinterpret_cast<const frag_t*>(base + QIMG + swz128((uint32_t)(wn * 64 + jn * 16 + j16), c));
#pragma unroll
            for (int i = 0; i < 8; i++)
#pragma unroll
                for (int jn = 0; jn < 4; jn++) acc[i][jn] = mma16<BF16>(av[i], wv[jn], acc[i][jn]);
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

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int jn = 0; jn < 4; jn++)
#pragma unroll
            for (int e = 0; e < 4; e++) acc[i][jn][e] = 0.f;

    // fragment addresses inside a slot: local row = (wave's 16-row block) * 16 + j, logical chunk 4 t + g (K step t = 0, 1)
    const int g4 = lane >> 4, j16 = lane & 15;
    uint32_t foff[2];
#pragma unroll
    for (int s = 0; s < 2; s++) foff[s] = swz128((uint32_t)j16, 4u * s + (uint32_t)g4);
    const uint32_t a_row0 = (uint32_t)wm * 64u * 128u;  // A slots: this wave row's 64 local rows (four 16-row blocks)
    const uint32_t b_row0 = (uint32_t)wn * 32u * 128u;  // B slots: this wave column's 32 local rows (two 16-row blocks)

    frag_t av[4][2], wv0[2][2], wv1[2][2];
    auto read_a = [&](const uint8_t* base, int u) {
#pragma unroll
        for (int ii = 0; ii < 4; ii++)
#pragma unroll
            for (int s = 0; s < 2; s++)
                av[ii][s] = *reinterpret_cast<const frag_t*>(base + u * PSLOT + a_row0 + ii * 2048 + foff[s]);
    };
    auto read_b = [&](const uint8_t* base, int u, frag_t (&wv)[2][2]) {
#pragma unroll
        for (int jj = 0; jj < 2; jj++)
#pragma unroll
            for (int s = 0; s < 2; s++) wv[jj][s] = *reinterpret_cast<const frag_t*>(base + u * PSLOT + b_row0 + jj * 2048 + foff[s]);
    };
    // one phase: (I0 / 2)-th half of the wave's rows (four 16-row blocks) x JN-th half of its columns (two 16-column blocks) x
    // both K steps = 16 MFMAs; every accumulator takes step 0 before step 1, like the two-stage loop
#define DCA_MMA8(I0, JN, WV)                                                                                          \
    do {                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
        __builtin_amdgcn_s_setprio(1);                                                                                \
        _Pragma("unroll") for (int s = 0; s < 2; s++) _Pragma("unroll") for (int ii = 0; ii < 4; ii++)                \
            _Pragma("unroll") for (int jj = 0; jj < 2; jj++)                                                          \
                acc[2 * (I0) + ii][2 * (JN) + jj] = mma16<BF16>(av[ii][s], WV[jj][s], acc[2 * (I0) + ii][2 * (JN) + jj]); \
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
// Variant 3 (round 5, the default): the ping-pong K loop of variant 2 with a LEAN layer tail for the network's own layer
// forms, relu(a . w^T + bias) and relu(a . w^T + bias + skip), on whole tiles.
// What the measurements said (profiles/r05_gemm16_*.txt; tools/gemm16_probe.py, tools/store_pattern_probe.hip):
//   * the tail was bound by its own INSTRUCTIONS, not by HBM: rounding to bf16 in software (six integer instructions per
//     value) alone cost 4.5 us of a 40 us tile — v_cvt_pk_bf16_f32 does two values per instruction (all variants now);
//     the accumulator-layout tail of variants 1 / 2 further issues 128 ds_write_b32 + 32 ds_read_b128 + 32 eight-byte
//     stores per wave, every one behind run-time tests for the ragged cases: ~4200 instructions per wave and tile;
//   * a store with one ROW per lane costs a CU 3.5 us per 128 KB tile however few CUs store at the time, one with 8 lanes
//     per 128-byte row segment 1.0 us.
// Three changes, same products in the same order (bit-identical to variants 1 / 2 — the race screen in tests/):
//   * OPERAND ROLES SWAPPED in the MFMA: the weight fragment is the instruction's A operand, the activation fragment its B
//     operand, so D[i][j] has j = lane & 31 = the activation ROW and i (the register index) = the output column.  Which
//     physical weight row feeds MFMA row i is free to choose (it is only the source address of an LDS-DMA piece): sigma16()
//     makes a lane's 8 consecutive accumulator registers 8 CONSECUTIVE output columns.  Bias, ReLU and the rounding then work
//     on register octets, and a lane leaves with packed 16-byte PIECES instead of 4-byte words.
//   * PIECE EXCHANGE: the four pieces a lane holds of its own row change hands inside the wave through a wave-private 4 KB
//     of LDS (4 ds_write_b128 + 4 ds_read_b128 per 32 rows) so that lane (4 g + x, h) stores piece x * 2 + h of rows
//     4 g .. 4 g + 3: 8 lanes per 128-byte row segment.  Skip rows come in by the same route, the other way round.
//   * NO RUN-TIME CASES in the tail: the kernel is compiled per form (skip or not, bias or not; ReLU always), takes whole
//     256 x 256 tiles with 16-byte aligned rows only, and the host hands everything else — other forms, the ragged right /
//     bottom strips of a layer — to variant 2.  ~520 instructions per wave and tile.
// Measured (204 800 rows, candidates taking turns, profiles/r05_gemm_bench.txt): K = 1024 0.450 / 0.499 ms (bias / residual
// form) against 0.491 / 0.529 for variant 2 and 0.376 / 0.562 for the library; K = 5120 1.93 / 1.98 against 1.92 / 1.97 and
// 1.67 / 1.88.
// Built on top of this, measured, and DELETED again in the same round: the PERSISTENT form the round-4 review asked for — one
// workgroup per CU walking its XCD's tiles, the exchange slices behind the operand slots (160 KB of LDS), the first seven
// half-tiles of the next tile requested before the tail so that they land under it.  Bit-identical, 0.454 / 0.500 ms at
// K = 1024 and 1.90 / 1.94 at K = 5120: nothing at K = 1024, 1.6 % at K = 5120 (profiles/r05_gemm16_timeline.txt).  Why: the
// prefetch (112 KB) and the tail's stores (128 KB) share the CU's one memory pipe, which moves ~15-20 B/clk with every CU
// active — the tail got 7.4 us long instead of disappearing — and the K loop itself runs at that same fetch rate (64 KB per
// K-tile in ~1.75 us), not at the matrix pipe's.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int sigma16(int jb, int i) {
    // 16 x 16 blocks: MFMA row i = 4 g + r of weight block jb (g = lane >> 4 of the lane that will hold it, r = its register)
    //   ->  column inside the wave's 64: (jb >> 1) * 32 + g * 8 + (jb & 1) * 4 + r,
    // so that lane (g, j) finds columns (jb >> 1) * 32 + g * 8 + 0..7 of ITS activation row in acc[.][2 p][0..3], acc[.][2 p + 1][0..3]
    return (jb >> 1) * 32 + (i >> 2) * 8 + (jb & 1) * 4 + (i & 3);
}


template <bool BF16>
__device__ __forceinline__ uint32_t pack16x2(float lo, float hi) {
    if constexpr (BF16)
        return pack_bf16x2(lo, hi);
    else
        return (uint32_t)to_f16(lo) | ((uint32_t)to_f16(hi) << 16);
}
template <bool BF16>
__device__ __forceinline__ float lo16(uint32_t v) {
    if constexpr (BF16)
        return __uint_as_float(v << 16);
    else
        return from_f16((uint16_t)(v & 0xFFFFu));
}
template <bool BF16>
__device__ __forceinline__ float hi16(uint32_t v) {
    if constexpr (BF16)
        return __uint_as_float(v & 0xFFFF0000u);
    else
        return from_f16((uint16_t)(v >> 16));
}

// Tail of a FULL tile.  tl: this wave's 4 KB exchange slice; rows cm0 + wm*128 .. +128, columns cn0 + wn*64 .. +64.
// Lane (g = lane >> 4, j = lane & 15) holds, of the 32-row group i: rows 16 b + j (b = 0, 1: accumulator blocks 2 i + b), of each
// the two 16-byte pieces c16 = 4 p + g (p = 0, 1: columns p * 32 + g * 8 .. + 8 = acc[2 i + b][2 p][0..3], acc[2 i + b][2 p + 1][0..3]).
template <bool BF16, bool SKIP, bool RELU, bool BIAS>
__device__ __forceinline__ void gemm16_tail_full(const Gemm16Args& p, uint8_t* tl, f32x4 (&acc)[8][4], int64_t cm0, int cn0, int wm,
                                                 int wn, int l31, int h) {
    const int tg = l31 >> 2, tx = l31 & 3;
    const int lane_ = h * 32 + l31, g4 = lane_ >> 4, j16 = lane_ & 15;
    // piece X = 2 b + p of the lane's four: row 16 b + j, c16 = 4 p + g
    auto own_addr = [&](int X) {
        const int row = 16 * (X >> 1) + j16, c16 = 4 * (X & 1) + g4;
        return tl + row * 128 + ((uint32_t)((c16 ^ (row & 7))) << 4);
    };
    auto quad_addr = [&](int j) {
        const int row = 4 * tg + j;
        return tl + row * 128 + ((uint32_t)(((tx * 2 + h) ^ (row & 7))) << 4);
    };
    const int colw = cn0 + wn * 64;
    if constexpr (BIAS) {  // bias of this lane's 2 x 8 columns, added in place
        const float* bp = p.bias + colw + g4 * 8;
#pragma unroll
        for (int pp = 0; pp < 2; pp++) {
            const float4 b0 = *reinterpret_cast<const float4*>(bp + pp * 32), b1 = *reinterpret_cast<const float4*>(bp + pp * 32 + 4);
            const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int ib = 0; ib < 8; ib++)
#pragma unroll
                for (int e = 0; e < 8; e++) acc[ib][2 * pp + (e >> 2)][e & 3] += bb[e];
            __builtin_amdgcn_sched_barrier(0);  // (left alone, the scheduler hoists every load of the tail to its top and spills)
        }
    }
    // this lane's pieces in the coalesced arrangement: piece tx * 2 + h of rows 4 tg + j of a 32-row block
    const int64_t rq = cm0 + wm * 128 + 4 * tg;
    const int64_t goff = rq * p.ldo + colw + (tx * 2 + h) * 8;
    u32x4 sk[2][4];
    auto load_skip = [&](int i, u32x4 (&s4)[4]) {
#pragma unroll
        for (int j = 0; j < 4; j++) s4[j] = *reinterpret_cast<const u32x4*>(p.skip + goff + (int64_t)(i * 32 + j) * p.ldo);
    };
    if constexpr (SKIP) load_skip(0, sk[0]);
    auto rows32 = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (SKIP) {
            if constexpr (i + 1 < 4) load_skip(i + 1, sk[(i + 1) & 1]);  // in flight while this block is worked on
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; j++) *reinterpret_cast<u32x4*>(quad_addr(j)) = sk[i & 1][j];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            u32x4 mine[4];
#pragma unroll
            for (int X = 0; X < 4; X++) mine[X] = *reinterpret_cast<const u32x4*>(own_addr(X));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int X = 0; X < 4; X++)
#pragma unroll
                for (int e = 0; e < 4; e++) {  // values 2 e, 2 e + 1 of the piece's eight: accumulator 2 p + (e >> 1), registers 2 (e & 1), + 1
                    acc[2 * i + (X >> 1)][2 * (X & 1) + (e >> 1)][2 * (e & 1)] += lo16<BF16>(mine[X][e]);
                    acc[2 * i + (X >> 1)][2 * (X & 1) + (e >> 1)][2 * (e & 1) + 1] += hi16<BF16>(mine[X][e]);
                }
        }
#pragma unroll
        for (int X = 0; X < 4; X++) {
            u32x4 ov;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float a = acc[2 * i + (X >> 1)][2 * (X & 1) + (e >> 1)][2 * (e & 1)],
                      b = acc[2 * i + (X >> 1)][2 * (X & 1) + (e >> 1)][2 * (e & 1) + 1];
                if constexpr (RELU) {
                    a = fmaxf(a, 0.f);
                    b = fmaxf(b, 0.f);
                }
                ov[e] = pack16x2<BF16>(a, b);
            }
            *reinterpret_cast<u32x4*>(own_addr(X)) = ov;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        u32x4 q[4];
#pragma unroll
        for (int j = 0; j < 4; j++) q[j] = *reinterpret_cast<const u32x4*>(quad_addr(j));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the slice is rewritten by the next block)
#pragma unroll
        for (int j = 0; j < 4; j++) *reinterpret_cast<u32x4*>(p.out + goff + (int64_t)(i * 32 + j) * p.ldo) = q[j];
        __builtin_amdgcn_sched_barrier(0);
    };
    rows32(std::integral_constant<int, 0>{});
    rows32(std::integral_constant<int, 1>{});
    rows32(std::integral_constant<int, 2>{});
    rows32(std::integral_constant<int, 3>{});
}

// (Round 6 ran this kernel with 16 / 12 fragment reads per K-tile instead of 24 — registers standing in for the skipped fragments,
// timing only — to price the one-wave-per-SIMD layout's only advantage: 2.5 % of the K loop, profiles/r06_gemm16_probe.txt.  The
// probe variants were removed when the kernels moved to the 16x16x32 instruction.)
template <bool BF16, bool SKIP, bool BIAS>
__global__ __launch_bounds__(QTHREADS, 2) void k_gemm16s(const Gemm16Args p) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    using frag_t = typename std::conditional<BF16, b16x8, h16x8>::type;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, l31 = lane & 31, h = lane >> 5;
    const int wm = w >> 2, wn = w & 3;
    const int nNt = p.n / QBN;  // (m, n: multiples of the tile)
    const int64_t slot = blockIdx.x >> 3;
    const int64_t m0 = ((slot / nNt) * 8 + (blockIdx.x & 7)) * QBM;  // the N tiles of one M tile sit on one XCD
    const int n0 = (int)(slot % nNt) * QBN;
    if (m0 >= p.m) return;
    const int nk = p.k / QBK;

    // (full tiles only — the host hands the ragged edges of a layer to variant 2: no clamping, no per-element tests)
    const uint16_t* src[4][2];
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const uint32_t r = (uint32_t)((q * 8 + w) * 8 + (lane >> 3));
            const uint32_t c = (uint32_t)(lane & 7) ^ ((r >> 1) & 7u);
            if (u < 2) {
                const int64_t gr = m0 + (r >> 6) * 128 + (u == PS_A23 ? 64 : 0) + (r & 63);
                src[u][q] = p.a + gr * p.lda + c * 8;
            } else {
                const int gn = n0 + (int)(r >> 5) * 64 + sigma16((u == PS_B1 ? 2 : 0) + (int)((r >> 4) & 1), (int)(r & 15));
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
    const int g4 = lane >> 4, j16 = lane & 15;
    uint32_t foff[2];
#pragma unroll
    for (int s = 0; s < 2; s++) foff[s] = swz128((uint32_t)j16, 4u * s + (uint32_t)g4);
    const uint32_t a_row0 = (uint32_t)wm * 64u * 128u;
    const uint32_t b_row0 = (uint32_t)wn * 32u * 128u;

    f32x4 acc[8][4];
    frag_t av[4][2], wv0[2][2], wv1[2][2];
    auto read_a = [&](const uint8_t* base, int u) {
#pragma unroll
        for (int ii = 0; ii < 4; ii++)
#pragma unroll
            for (int s = 0; s < 2; s++)
                av[ii][s] = *reinterpret_cast<const frag_t*>(base + u * PSLOT + a_row0 + ii * 2048 + foff[s]);
    };
    auto read_b = [&](const uint8_t* base, int u, frag_t (&wv)[2][2]) {
#pragma unroll
        for (int jj = 0; jj < 2; jj++)
#pragma unroll
            for (int s = 0; s < 2; s++) wv[jj][s] = *reinterpret_cast<const frag_t*>(base + u * PSLOT + b_row0 + jj * 2048 + foff[s]);
    };
    // (weight fragment = A operand, activation fragment = B operand: see the header)
#define DCA_MMA8(I0, JN, WV)                                                                                          \
    do {                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
        __builtin_amdgcn_s_setprio(1);                                                                                \
        _Pragma("unroll") for (int s = 0; s < 2; s++) _Pragma("unroll") for (int ii = 0; ii < 4; ii++)                \
            _Pragma("unroll") for (int jj = 0; jj < 2; jj++)                                                          \
                acc[2 * (I0) + ii][2 * (JN) + jj] = mma16<BF16>(WV[jj][s], av[ii][s], acc[2 * (I0) + ii][2 * (JN) + jj]); \
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

#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int jn = 0; jn < 4; jn++)
#pragma unroll
            for (int e = 0; e < 4; e++) acc[i][jn][e] = 0.f;
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
    if (wm == 0) DCA_BAR();  // ... and the first waits for it here: nobody reads the operand slots any more
    // the tail's piece exchange uses the first 32 KB of the (now idle) operand slots, 4 KB per wave
    gemm16_tail_full<BF16, SKIP, true, BIAS>(p, lds + w * 4096, acc, m0, n0, wm, wn, l31, h);
#undef DCA_VMCNT
#undef DCA_MMA8
}

}  // namespace dca

using namespace dca;

static int g_gemm16_variant = 3;

// the lean-tail kernel serving one of the network's layer forms: relu(a . w^T (+ bias) (+ skip)) — FastResnet folds the second
// bias of a residual block into the weights (a constant-one input unit), so its residual layers come without a bias
static const void* lean_kernel(int bf16, int skip, int bias) {
    static const void* tab[8] = {
        reinterpret_cast<const void*>(k_gemm16s<false, false, false>), reinterpret_cast<const void*>(k_gemm16s<true, false, false>),
        reinterpret_cast<const void*>(k_gemm16s<false, true, false>),  reinterpret_cast<const void*>(k_gemm16s<true, true, false>),
        reinterpret_cast<const void*>(k_gemm16s<false, false, true>),  reinterpret_cast<const void*>(k_gemm16s<true, false, true>),
        reinterpret_cast<const void*>(k_gemm16s<false, true, true>),   reinterpret_cast<const void*>(k_gemm16s<true, true, true>)};
    return tab[(bf16 ? 1 : 0) | (skip ? 2 : 0) | (bias ? 4 : 0)];
}

extern "C" {

/* test hook: 1 = two K-step stages, one drain + barrier per K-step (the reference the race screens compare against); 2 = the
 * 8-phase ping-pong schedule with the general tail (what ragged strips and other layer forms always run on); 3 (default) = the
 * same schedule with operand roles swapped and the lean tail for the network's layer forms */
int dca_gemm16_variant(int v) {
    DCA_ARG(v >= 1 && v <= 3);
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
            for (int f = 0; f < 8; f++) DCA_HIP(hipFuncSetAttribute(lean_kernel(f & 1, (f >> 1) & 1, f >> 2), hipFuncAttributeMaxDynamicSharedMemorySize, QLDS));
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
    hipStream_t s = (hipStream_t)stream;
    // one workgroup per 256 x 256 tile, any shape, any form (variants 1, 2)
    auto launch_generic = [&](const Gemm16Args& q, int variant) -> int {
        const int64_t nMt = (q.m + QBM - 1) / QBM;
        const int64_t nNt = (q.n + QBN - 1) / QBN;
        const int64_t blocks = ((nMt + 7) / 8) * 8 * nNt;
        if (blocks > 0x7FFFFFFFll) {
            set_error("dca_gemm16: too many tiles");
            return DCA_E_BADARG;
        }
        const dim3 grid((unsigned)blocks), block(QTHREADS);
        if (variant == 1) {
            if (dtype == DCA_DT_BF16)
                hipLaunchKernelGGL(k_gemm16<true>, grid, block, QLDS, s, q);
            else
                hipLaunchKernelGGL(k_gemm16<false>, grid, block, QLDS, s, q);
        } else {
            if (dtype == DCA_DT_BF16)
                hipLaunchKernelGGL(k_gemm16p<true>, grid, block, QLDS, s, q);
            else
                hipLaunchKernelGGL(k_gemm16p<false>, grid, block, QLDS, s, q);
        }
        return 0;
    };
    if (g_gemm16_variant < 3) {
        if (int rc = launch_generic(p, g_gemm16_variant)) return rc;
        return launch_check("k_gemm16");
    }
    // Variant 3 — the lean-tail kernel — serves the network's own layer forms, relu(a . w^T (+ bias) (+ skip)), on whole tiles
    // with 16-byte aligned rows; any other form goes to variant 2, and so do the ragged right and bottom strips of a layer
    // (none in the network's own shapes: widths are padded to 1024 / 5120, row counts to 1024) — the same products in the same
    // order either way.
    const bool lean = relu && (ldo % 8 == 0) && (((uintptr_t)out | (uintptr_t)skip | (uintptr_t)bias) % 16 == 0);
    const int64_t mf = lean ? (m / QBM) * QBM : 0;
    const int nf = lean ? (n / QBN) * QBN : 0;
    if (mf > 0 && nf > 0) {
        Gemm16Args q = p;
        q.m = mf;
        q.n = nf;
        const int64_t nMt = mf / QBM, nNt = nf / QBN;
        const int64_t blocks = ((nMt + 7) / 8) * 8 * nNt;
        if (blocks > 0x7FFFFFFFll) {
            set_error("dca_gemm16: too many tiles");
            return DCA_E_BADARG;
        }
        void* kargs[] = {&q};
        DCA_HIP(hipLaunchKernel(lean_kernel(dtype == DCA_DT_BF16, skip != nullptr, bias != nullptr), dim3((unsigned)blocks), dim3(QTHREADS), kargs, QLDS, s));
    }
    auto strip = [&](int64_t r0, int64_t rows, int c0, int cols) -> int {  // rows [r0, r0 + rows) x columns [c0, c0 + cols)
        if (rows <= 0 || cols <= 0) return 0;
        Gemm16Args q = p;
        q.a = p.a + r0 * lda;
        q.w = p.w + (int64_t)c0 * ldw;
        q.bias = p.bias ? p.bias + c0 : nullptr;
        q.skip = p.skip ? p.skip + r0 * ldo + c0 : nullptr;
        q.out = p.out + r0 * ldo + c0;
        q.m = rows;
        q.n = cols;
        return launch_generic(q, 2);
    };
    if (int rc = strip(0, mf, nf, n - nf)) return rc;  // right strip beside the full tiles
    if (int rc = strip(mf, m - mf, 0, n)) return rc;   // bottom strip, full width
    return launch_check("k_gemm16s");
}

}  // extern "C"
