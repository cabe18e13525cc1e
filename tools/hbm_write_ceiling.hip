// Measures the pure HBM write ceiling on this GPU (16-B stores, plain vs nontemporal), to put the fused
// expansion kernel's 5.7 TB/s in context.  Build: hipcc --offload-arch=gfx950 -O3 tools/hbm_write_ceiling.hip -o /tmp/hwc
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int NT>
__global__ void wr(u32x4* p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    u32x4 v = u32x4{(unsigned)i, 1u, 2u, 3u};
    for (; i < n; i += stride) {
        if (NT) __builtin_nontemporal_store(v, &p[i]); else p[i] = v;
    }
}
// tile-contiguous like the expansion kernel: block b writes its own contiguous 1 MB region
template <int NT>
__global__ void wr_tile(u32x4* p, size_t chunks_per_block) {
    u32x4* q = p + (size_t)blockIdx.x * chunks_per_block;
    u32x4 v = u32x4{blockIdx.x, 1u, 2u, 3u};
    for (size_t i = threadIdx.x; i < chunks_per_block; i += blockDim.x) {
        if (NT) __builtin_nontemporal_store(v, &q[i]); else q[i] = v;
    }
}
__global__ void cp(const u32x4* __restrict__ a, u32x4* __restrict__ b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) b[i] = a[i];
}
int main() {
    size_t bytes = 16ull << 30, n = bytes / 16;
    u32x4 *p, *q;
    hipMalloc(&p, bytes); hipMalloc(&q, bytes / 2);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto time = [&](const char* name, auto f, double gb) {
        f(); hipDeviceSynchronize();
        hipEventRecord(a); for (int i = 0; i < 5; i++) f(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
        printf("%-28s %8.3f ms  %8.1f GB/s\n", name, ms, gb / (ms * 1e-3));
    };
    double gb = bytes / 1e9;
    for (int blocks : {2048, 8192, 65536})
        time(blocks == 2048 ? "plain grid2048" : blocks == 8192 ? "plain grid8192" : "plain grid65536",
             [&] { hipLaunchKernelGGL(wr<0>, dim3(blocks), dim3(256), 0, 0, p, n); }, gb);
    time("nontemporal grid8192", [&] { hipLaunchKernelGGL(wr<1>, dim3(8192), dim3(256), 0, 0, p, n); }, gb);
    size_t cpb = (1 << 20) / 16;  // 1 MB per block
    time("tile 1MB/block plain", [&] { hipLaunchKernelGGL(wr_tile<0>, dim3(n / cpb), dim3(256), 0, 0, p, cpb); }, gb);
    time("tile 1MB/block nontemporal", [&] { hipLaunchKernelGGL(wr_tile<1>, dim3(n / cpb), dim3(256), 0, 0, p, cpb); }, gb);
    time("copy 8GB->8GB (r+w)", [&] { hipLaunchKernelGGL(cp, dim3(8192), dim3(256), 0, 0, p, q, n / 2); }, gb);
    return 0;
}
