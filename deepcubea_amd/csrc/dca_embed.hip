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
//   * a workgroup (16 waves; 12 where the weights leave no room for more) owns NT output columns; their fp32 weights, transposed
//     — [K][NT], one 4 * NT-byte row per one-hot column — sit in LDS for the workgroup's whole life (NT = 64: cube3 83 KB, puzzle15
//     66 KB, puzzle24 160 KB; NT = 16: puzzle35 83 KB, puzzle48 154 KB); after staging them the workgroup never meets at a barrier again;
//   * each WAVE walks over steps of 64 / (NT / 4) * T states on its own: the step's state bytes come in as one 16-byte load per lane
//     (issued three steps ahead) into a wave-private LDS slice; a lane owns (state, 4 consecutive columns): it pulls its state's
//     bytes out of the slice as aligned dwords + v_alignbyte, then per position one row address, one ds_read_b128 and four adds —
//     positions in ascending order, so a state's value has the same bits in any batch;
//   * with NT = 64 a table row is 256 bytes, so the row address of state byte s is `lane offset | s << 8` — ONE SDWA byte move
//     (v_mov_b32_sdwa dst_sel:BYTE_1) into a register that keeps the lane offset, instead of v_bfe + v_lshl_add: 3 vector
//     instructions per position instead of 4.4 (puzzle15 2.44 -> 2.11 ms, puzzle24 3.97 -> 3.68);
//   * the NT / 4 lanes of a state read 4 * NT contiguous bytes of one LDS row: for NT = 64 the lane groups of a ds_read_b128
//     touch disjoint banks (SQ_LDS_BANK_CONFLICT = 0); for NT = 16 four states share a lane group and their random rows collide
//     (46 % of the LDS cycles, profiles/r06_l1_embed_pmc.txt) — the price of the 2401-row table;
//   * the tail stores 2 * NT contiguous bytes per state and plane (fp16 planes for the f16x3 layers, bf16, fp32, or e4m3 bytes
//     for the fp8 layers); the blockIdx -> (column tile, row slice) map hands each XCD consecutive column tiles of one row slice,
//     so the pieces of an output line meet in one L2 (puzzle35 planes 8.2 -> 5.1 ms).
// What bounds it (rocprofv3 PMC, tools/valu_rate_probe.hip): with NT = 64 the vector ALU (two v_pk_add_f32 of ~4.9 cycles and the
// address per position) with the LDS array half busy; with NT = 16 the LDS array, 84 % busy, 46 % of it conflict cycles.
// Built, measured, removed (profiles/r06_l1_embed_experiments.txt): a "position-group" form for puzzle35 / 48 — 64-column tiles,
// the accumulators of 640-1024 states in registers, the table streamed past them in LDS-DMA slices between barriers: no bank
// conflicts, but the restaging and the per-slice state loads made it 1.5-2.2x SLOWER (puzzle48 9.5-14.2 ms against 6.5).
// Per 409 600 rows x 5120 units (profiles/r06_l1_embed_bench.txt), one-hot MFMA kernel -> this kernel, fp16 planes out:
// puzzle15 3.0 -> 2.1 ms, puzzle24 9.2 -> 3.7, puzzle35 17.3 -> 5.2, puzzle48 30.1 -> 6.5; cube3 3.9 -> 6.0 (stays on MFMA).
#include "dca_common.h"

namespace dca {

template <int D, int DEPTH, int NT, int WAVES, int T>
struct EmbGeo {
    static constexpr int K = D * DEPTH;
    static constexpr int LPS = NT / 4;                                // lanes per state (4 columns each)
    static constexpr int SPW = 64 / LPS * T;                          // states per wave-step (T per lane)
    static constexpr int W_BYTES = K * NT * 4;
    static constexpr int NPIECE = (SPW * D + 15 + 15) / 16;           // 16-byte pieces covering a wave-step's rows from the 16-byte boundary below them
    static constexpr int SLICE = NPIECE * 16 + 16;                    // (+ slack: the dword reads run past the last row)
    static constexpr int LDS = W_BYTES + NT * 4 + WAVES * SLICE;
    static constexpr int NW = (D + 3) / 4 + 1;                        // aligned dwords covering one state row at any byte offset
    static_assert(NPIECE <= 64, "one 16-byte piece per lane");
};

// relu, conversion and store of one lane's 4 consecutive columns of row r (OUT: 0 fp32, 2 bf16, 4 two fp16 planes — high, then low at
// + m * n_pad —, 5 e4m3 saturating); returns whether a value left the fp16 range (planes only)
template <int OUT>
__device__ __forceinline__ bool emb_store(const float4& acc, int relu, int64_t r, int64_t m, int64_t n_pad, int64_t col, void* __restrict__ out) {
    float u[4] = {acc.x, acc.y, acc.z, acc.w};
    if (relu) {
#pragma unroll
        for (int e = 0; e < 4; e++) u[e] = fmaxf(u[e], 0.f);
    }
    bool ovf = false;
    if (r < m) {
        const int64_t o = r * n_pad + col;
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
    return ovf;
}

// byte 1 of `a` <- byte B of `w`, the other bytes of `a` kept: with 256-byte table rows (NT = 64) `a` = lane offset (byte 0) | window
// (bytes 2-3) becomes the row address of state byte s in ONE vector instruction (v_bfe + v_lshl_add otherwise)
template <int B>
__device__ __forceinline__ void put_byte1(uint32_t& a, uint32_t w) {
    if constexpr (B == 0) asm("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0" : "+v"(a) : "v"(w));
    if constexpr (B == 1) asm("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1" : "+v"(a) : "v"(w));
    if constexpr (B == 2) asm("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2" : "+v"(a) : "v"(w));
    if constexpr (B == 3) asm("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3" : "+v"(a) : "v"(w));
}

template <int DEPTH, int POS, int N>
struct EmbSum {  // positions POS .. N - 1 of one state, unrolled at compile time (the byte selector is an instruction field)
    template <int NWIN, int NV>
    static __device__ __forceinline__ void run(float4& acc, uint32_t (&ab)[NWIN][4], const uint32_t (&v)[NV]) {
        if constexpr (POS < N) {
            constexpr int row0 = POS * DEPTH, win = row0 >> 8, r = row0 & 255;
            uint32_t& a = ab[win][POS & 3];
            put_byte1<POS & 3>(a, v[POS >> 2]);
            // `a` is the LDS address itself (the table starts at LDS offset 0: checked at kernel entry)
            typedef float f4v __attribute__((ext_vector_type(4)));
            const f4v gw = *reinterpret_cast<const __attribute__((address_space(3))) f4v*>((uintptr_t)(a + (uint32_t)(r * 256)));
            acc.x += gw.x;
            acc.y += gw.y;
            acc.z += gw.z;
            acc.w += gw.w;
            EmbSum<DEPTH, POS + 1, N>::run(acc, ab, v);
        }
    }
};

template <int D, int DEPTH, int NT, int WAVES, int T, int OUT /*0 fp32, 2 bf16, 4 two fp16 planes (high, then low at + m * ldo), 5 e4m3 (saturating)*/,
          bool SD = false /*row addresses by SDWA byte moves (256-byte table rows: NT = 64)*/>
__global__ __launch_bounds__(WAVES * 64) void k_l1_embed(const uint8_t* __restrict__ nn, int64_t m, const float* __restrict__ wt /*[K][n_pad]*/,
                                                         int64_t n_pad, const float* __restrict__ bias, int relu, void* __restrict__ out,
                                                         int* __restrict__ overflow) {
    using G = EmbGeo<D, DEPTH, NT, WAVES, T>;
    extern __shared__ __attribute__((aligned(16))) uint8_t le[];
    float* lw = reinterpret_cast<float*>(le);
    float* lb = lw + G::K * NT;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    if constexpr (SD) {  // the SDWA row addresses are absolute LDS addresses: the weight table must start at LDS offset 0
        if ((uint32_t)(uintptr_t)((__attribute__((address_space(3))) uint8_t*)le) != 0u) __builtin_trap();
    }
    uint8_t* ls = le + G::W_BYTES + NT * 4 + wave * G::SLICE;  // this wave's slice: no other wave touches it, no workgroup barrier in the loop
    // workgroups go to the 8 XCDs round-robin in launch order: hand each XCD a run of CONSECUTIVE column tiles of one row slice, so
    // that the pieces of an output line written by neighbouring tiles meet in one L2 instead of reaching memory one by one
    uint32_t bx = blockIdx.x, by = blockIdx.y;
    {
        const uint32_t nwg = gridDim.x * gridDim.y, lin = blockIdx.x + blockIdx.y * gridDim.x;
        if ((nwg & 7u) == 0) {
            const uint32_t v = (lin & 7u) * (nwg >> 3) + (lin >> 3);
            by = v / gridDim.x;
            bx = v - by * gridDim.x;
        }
    }
    const int64_t n0 = (int64_t)bx * NT;
    const int64_t nsteps = (m + G::SPW - 1) / G::SPW, total = m * D;
    const int64_t gstride = (int64_t)gridDim.y * WAVES;

    // a wave-step's rows start at byte g * SPW * D of the matrix; the wave fetches from the 16-byte boundary below (the matrix
    // itself is 16-byte aligned) — one piece per lane, nothing read past the matrix's end (its last piece byte by byte)
    auto prefetch = [&](int64_t g) -> uint4 {
        uint4 v = make_uint4(0u, 0u, 0u, 0u);  // (zeros past the matrix: a zero byte is a valid colour)
        const int64_t q = ((g * (G::SPW * D)) & ~(int64_t)15) + lane * 16;
        if (lane < G::NPIECE && g < nsteps) {
            if (q + 16 <= total) {
                v = *reinterpret_cast<const uint4*>(nn + q);
            } else if (q < total) {
                uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                for (int e = 0; e < 16; e++)
                    if (q + e < total) w[e >> 2] |= (uint32_t)nn[q + e] << (8 * (e & 3));
                v = make_uint4(w[0], w[1], w[2], w[3]);
            }
        }
        return v;
    };
    int64_t g = (int64_t)by * WAVES + wave;
    // three steps ahead: the counter a wave waits on (vmcnt) retires loads and stores in issue order, so a load issued one step
    // ahead would make every step wait for the previous step's output stores to reach memory
    uint4 pre = prefetch(g), pre1 = prefetch(g + gstride), pre2 = prefetch(g + 2 * gstride);
    // the tile's weights (row k = one-hot column k: NT floats of the transposed matrix) and its bias
    for (int q = t; q < G::K * (NT / 4); q += WAVES * 64) {
        const int k = q / (NT / 4), c = q - k * (NT / 4);
        reinterpret_cast<float4*>(lw)[q] = *reinterpret_cast<const float4*>(wt + (int64_t)k * n_pad + n0 + 4 * c);
    }
    if (t < NT) lb[t] = bias[n0 + t];
    __syncthreads();
    const int sl = lane / G::LPS, cp = lane - sl * G::LPS;
    const uint8_t* wcol = reinterpret_cast<const uint8_t*>(lw) + cp * 16;
    bool ovf = false;
    for (; g < nsteps; g += gstride) {
        if (lane < G::NPIECE) *reinterpret_cast<uint4*>(ls + lane * 16) = pre;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private slice: only the wave's own writes
        __builtin_amdgcn_wave_barrier();
        pre = pre1;
        pre1 = pre2;
        pre2 = prefetch(g + 3 * gstride);
        for (int j = 0; j < T; j++) {  // the lane's T states of this step (state j * 64 / LPS + sl of the step)
            // the state's D bytes from the image: aligned dwords, shifted into place
            const int sj = j * (64 / G::LPS) + sl;
            const uint32_t boff = (uint32_t)((g * (G::SPW * D)) & 15) + (uint32_t)sj * D, sh = boff & 3u;
            const uint32_t* wsrc = reinterpret_cast<const uint32_t*>(ls + (boff & ~3u));
            uint32_t w[G::NW];
#pragma unroll
            for (int i = 0; i < G::NW; i++) w[i] = wsrc[i];
            uint32_t v[G::NW - 1];
#pragma unroll
            for (int i = 0; i < G::NW - 1; i++) v[i] = __builtin_amdgcn_alignbyte(w[i + 1], w[i], sh);
            float4 acc = *reinterpret_cast<const float4*>(lb + 4 * cp);
            if constexpr (SD) {
                static_assert(NT == 64, "SDWA row addresses need 256-byte rows");
                constexpr int NWIN = (((D - 1) * DEPTH) >> 8) + 1;
                uint32_t ab[NWIN][4];
#pragma unroll
                for (int wi = 0; wi < NWIN; wi++)
#pragma unroll
                    for (int k = 0; k < 4; k++) ab[wi][k] = (uint32_t)cp * 16u + (uint32_t)wi * 65536u;
                EmbSum<DEPTH, 0, D>::run(acc, ab, v);
            } else {
#pragma unroll
                for (int pos = 0; pos < D; pos++) {
                    const uint32_t s = (v[pos >> 2] >> (8 * (pos & 3))) & 0xFFu;
                    const float4 gw = *reinterpret_cast<const float4*>(wcol + (s + (uint32_t)(pos * DEPTH)) * (uint32_t)(NT * 4));
                    acc.x += gw.x;
                    acc.y += gw.y;
                    acc.z += gw.z;
                    acc.w += gw.w;
                }
            }
            ovf |= emb_store<OUT>(acc, relu, g * G::SPW + sj, m, n_pad, n0 + 4 * cp, out);
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();  // (every lane has read its rows: the slice may be rewritten by the next step)
    }
    if (OUT == 4 && ovf && overflow) *overflow = 1;
}

template <int D, int DEPTH, int NT, int WAVES, int T>
int launch_embed(const uint8_t* nn, int64_t m, const float* wt, int64_t n_pad, const float* bias, int relu, void* out, int out_dtype,
                 int* overflow, hipStream_t s) {
    using G = EmbGeo<D, DEPTH, NT, WAVES, T>;
    static_assert(G::LDS <= 160 * 1024, "weight slice does not fit LDS");
    static_assert(NT % 4 == 0 && 64 % (NT / 4) == 0, "tile geometry");
    if (n_pad % NT != 0) {
        set_error("dca_l1_embed: n_pad %lld is not a multiple of the column tile %d", (long long)n_pad, NT);
        return DCA_E_BADARG;
    }
    const int64_t steps = (m + G::SPW - 1) / G::SPW, tiles = n_pad / NT;
    int64_t gy = (1024 + tiles - 1) / tiles;  // ~4 workgroups per CU over the launch (the weight slice is staged once per workgroup)
    if (gy * WAVES > steps) gy = (steps + WAVES - 1) / WAVES;
    if (gy < 1) gy = 1;
    const dim3 grid((unsigned)tiles, (unsigned)gy), block(WAVES * 64);
#define DCA_EMB_LAUNCH(OUTV)                                                                                            \
    do {                                                                                                                \
        auto kern = k_l1_embed<D, DEPTH, NT, WAVES, T, OUTV, NT == 64>;                                                 \
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
    if (state_dim == 54) return launch_embed<54, 6, 64, 16, 4>(nnet_in, m, w_t, n_pad, bias, relu, out, out_dtype, overflow, s);
    if (state_dim == 16) return launch_embed<16, 16, 64, 16, 8>(nnet_in, m, w_t, n_pad, bias, relu, out, out_dtype, overflow, s);
    if (state_dim == 25) return launch_embed<25, 25, 64, 12, 2>(nnet_in, m, w_t, n_pad, bias, relu, out, out_dtype, overflow, s);
    if (state_dim == 36) return launch_embed<36, 36, 16, 16, 1>(nnet_in, m, w_t, n_pad, bias, relu, out, out_dtype, overflow, s);
    if (depth == 49) return launch_embed<49, 49, 16, 12, 1>(nnet_in, m, w_t, n_pad, bias, relu, out, out_dtype, overflow, s);
    return launch_embed<49, 6, 64, 16, 4>(nnet_in, m, w_t, n_pad, bias, relu, out, out_dtype, overflow, s);
}

}  // extern "C"
