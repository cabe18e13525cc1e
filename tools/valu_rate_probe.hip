// What a wave64 VALU instruction costs on gfx950, for the instructions the embedding-sum kernel (csrc/dca_embed.hip) is made of:
// v_add_f32, v_pk_add_f32, v_pk_fma_f32, v_lshl_add_u32, v_bfe_u32 — 8 independent chains per wave, W waves per SIMD, every CU busy.
// Prints cycles per wave-instruction per SIMD (4 = full rate: 16 lanes per clock).
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate_probe.hip -o tools/bin/valu_rate_probe && tools/bin/valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int ITERS = 65536;

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, float seed) {
    f2 a[8];
    unsigned u[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        a[i] = f2{seed + i, seed * 2 + i};
        u[i] = (unsigned)(seed) + i + threadIdx.x;
    }
    const f2 b = {seed * 0.5f, seed * 0.25f};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if constexpr (OP == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(b.x));
            if constexpr (OP == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if constexpr (OP == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
            if constexpr (OP == 3) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
            if constexpr (OP == 4) asm volatile("v_bfe_u32 %0, %0, 1, 31" : "+v"(u[i]));
            if constexpr (OP == 5) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i].x + a[i].y + (float)u[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
void run(const char* name, int wps) {
    const int blocks = 256 * wps;  // 4 waves per block = one per SIMD; wps blocks per CU
    float* out;
    long long* cyc;
    hipMalloc(&out, blocks * 256 * 4);
    hipMalloc(&cyc, blocks * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(out, cyc, 1.5f);
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(out, cyc, 1.5f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // wall clock: wave-instructions per SIMD = wps * ITERS * 8
    const double inst = (double)wps * ITERS * 8;
    printf("%-16s %d waves/SIMD: %.3f ms -> %.2f ns per wave-instruction per SIMD (= %.2f cycles at 2.4 GHz)\n", name, wps, ms,
           ms * 1e6 / inst, ms * 1e6 / inst * 2.4);
    hipFree(out);
    hipFree(cyc);
}

int main() {
    for (int wps : {1, 2, 4}) {
        run<0>("v_add_f32", wps);
        run<1>("v_pk_add_f32", wps);
        run<2>("v_pk_fma_f32", wps);
        run<3>("v_lshl_add_u32", wps);
        run<4>("v_bfe_u32", wps);
        run<5>("v_and_or_b32", wps);
    }
    return 0;
}
