// What does a bf16 MFMA cost on RANDOM data?  hipBLASLt's gfx950 bf16 kernel is built on the 16x16x32 instruction (its name
// says MT256x256x64_MI16x16x1), the hand-written dca_gemm16 on 32x32x16 — and on random operands ours loses 26 % to the power
// limit where the library loses 7 % (profiles/r06_gemm16_probe.txt), whatever the number of LDS reads.  This probe issues
// nothing but MFMAs from registers — 8 waves per CU, 128 accumulator registers per wave like the real kernel — with either
// shape, on random and on all-zero operands, and prints the sustained TFLOP/s.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_power_probe.hip -o tools/bin/mfma_power_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 b16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>  // 32: v_mfma_f32_32x32x16_bf16, 8 accumulators of 16 registers; 16: v_mfma_f32_16x16x32_bf16, 32 of 4
__global__ __launch_bounds__(512, 2) void k_mfma(const b16x8* __restrict__ ops, float* __restrict__ out, int iters) {
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    b16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        a[i] = ops[(size_t)t * 8 + i];
        b[i] = ops[(size_t)t * 8 + 4 + i];
    }
    float s = 0.f;
    if constexpr (SHAPE == 32) {
        f32x16 acc[8];
#pragma unroll
        for (int j = 0; j < 8; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[j][e] = 0.f;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int kk = 0; kk < 4; kk++)  // one K-tile of 64 of the real kernel: 4 slices x 8 blocks = 32 MFMAs
#pragma unroll
                for (int j = 0; j < 8; j++) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(j + kk) & 3], b[(j >> 1) & 3], acc[j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 8; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) s += acc[j][e];
    } else {
        f32x4 acc[32];
#pragma unroll
        for (int j = 0; j < 32; j++)
#pragma unroll
            for (int e = 0; e < 4; e++) acc[j][e] = 0.f;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int kk = 0; kk < 2; kk++)  // the same flops per iteration: 2 slices of 32 x 32 blocks of 16 x 16 = 64 MFMAs
#pragma unroll
                for (int j = 0; j < 32; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(j + kk) & 3], b[(j >> 2) & 3], acc[j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 32; j++)
#pragma unroll
            for (int e = 0; e < 4; e++) s += acc[j][e];
    }
    out[t] = s;
}

typedef int i32x8 __attribute__((ext_vector_type(8)));
template <int SHAPE>  // e4m3 operands: 32: v_mfma_f32_32x32x64_f8f6f4, 8 accumulators of 16 registers; 16: v_mfma_f32_16x16x128_f8f6f4, 32 of 4
__global__ __launch_bounds__(512, 2) void k_mfma8(const i32x8* __restrict__ ops, float* __restrict__ out, int iters) {
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    i32x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        a[i] = ops[(size_t)t * 4 + (i & 1)];
        b[i] = ops[(size_t)t * 4 + 2 + (i & 1)];
        a[i][0] ^= i;  // (four different fragments from two loads)
        b[i][1] ^= i;
    }
    float s = 0.f;
    if constexpr (SHAPE == 32) {
        f32x16 acc[8];
#pragma unroll
        for (int j = 0; j < 8; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[j][e] = 0.f;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int kk = 0; kk < 2; kk++)
#pragma unroll
                for (int j = 0; j < 8; j++)
                    acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[(j + kk) & 3], b[(j >> 1) & 3], acc[j], 0, 0, 0, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 8; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) s += acc[j][e];
    } else {
        f32x4 acc[32];
#pragma unroll
        for (int j = 0; j < 32; j++)
#pragma unroll
            for (int e = 0; e < 4; e++) acc[j][e] = 0.f;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int j = 0; j < 32; j++)
                acc[j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[j & 3], b[(j >> 2) & 3], acc[j], 0, 0, 0, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 32; j++)
#pragma unroll
            for (int e = 0; e < 4; e++) s += acc[j][e];
    }
    out[t] = s;
}

int main(int argc, char** argv) {
    const int blocks = 256, threads = 512, iters = argc > 1 ? atoi(argv[1]) : 4000;
    const size_t n = (size_t)blocks * threads * 8;  // b16x8 per thread: 8
    uint16_t* h = (uint16_t*)malloc(n * 16);
    b16x8* d;
    float* o;
    hipMalloc(&d, n * 16);
    hipMalloc(&o, (size_t)blocks * threads * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int fill = 0; fill < 2; fill++) {
        uint64_t st = 88172645463325252ull;
        for (size_t i = 0; i < n * 8; i++) {
            st ^= st << 13; st ^= st >> 7; st ^= st << 17;
            // random bf16 in (-2, 2): sign, exponent 120..127, 7 random mantissa bits — what activations / weights look like
            h[i] = fill ? 0 : (uint16_t)(((st >> 20) & 0x8000u) | ((120 + ((st >> 40) & 7)) << 7) | ((st >> 50) & 0x7Fu));
        }
        hipMemcpy(d, h, n * 16, hipMemcpyHostToDevice);
        for (int round = 0; round < 3; round++)
            for (int shape = 0; shape < 2; shape++) {
                hipEventRecord(e0);
                for (int rep = 0; rep < 5; rep++) {
                    if (shape == 0)
                        hipLaunchKernelGGL(k_mfma<32>, dim3(blocks), dim3(threads), 0, 0, d, o, iters);
                    else
                        hipLaunchKernelGGL(k_mfma<16>, dim3(blocks), dim3(threads), 0, 0, d, o, iters);
                }
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                ms /= 5;
                const double flops = (double)blocks * 8 /*waves*/ * iters * 32.0 * 32768.0;  // 32 x (32x32x16) per wave and iteration either way
                if (round > 0)
                    printf("%-7s %-28s %8.3f ms  %8.1f TFLOP/s\n", fill ? "zeros" : "random",
                           shape == 0 ? "v_mfma_f32_32x32x16_bf16" : "v_mfma_f32_16x16x32_bf16", ms, flops / ms / 1e9);
            }
    }
    // ---- e4m3: 16 x (32x32x64) per wave and iteration either way
    for (int fill = 0; fill < 2; fill++) {
        uint64_t st = 88172645463325252ull;
        uint8_t* hb = (uint8_t*)h;
        for (size_t i = 0; i < n * 16; i++) {
            st ^= st << 13; st ^= st >> 7; st ^= st << 17;
            // random e4m3 in (-2, 2): sign, exponent 4..7, 3 random mantissa bits
            hb[i] = fill ? 0 : (uint8_t)(((st >> 20) & 0x80u) | ((4 + ((st >> 40) & 3)) << 3) | ((st >> 50) & 7u));
        }
        hipMemcpy(d, h, n * 16, hipMemcpyHostToDevice);
        for (int round = 0; round < 3; round++)
            for (int shape = 0; shape < 2; shape++) {
                hipEventRecord(e0);
                for (int rep = 0; rep < 5; rep++) {
                    if (shape == 0)
                        hipLaunchKernelGGL(k_mfma8<32>, dim3(blocks), dim3(threads), 0, 0, (const i32x8*)d, o, iters);
                    else
                        hipLaunchKernelGGL(k_mfma8<16>, dim3(blocks), dim3(threads), 0, 0, (const i32x8*)d, o, iters);
                }
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                ms /= 5;
                const double flops = (double)blocks * 8 * iters * 16.0 * 131072.0;
                if (round > 0)
                    printf("%-7s %-28s %8.3f ms  %8.1f TFLOP/s\n", fill ? "zeros" : "random",
                           shape == 0 ? "v_mfma_f32_32x32x64_f8f6f4" : "v_mfma_f32_16x16x128_f8f6f4", ms, flops / ms / 1e9);
            }
    }
    return 0;
}
