// What does a GEMM tile's output store cost one CU?  One 512-thread workgroup per CU writes a 256 x 256 bf16 tile (128 KB,
// row stride 2 KB: a 1024-column activation) with 16-byte stores in three lane -> address patterns, timed per workgroup on
// the device wall clock, with `active` of the 256 workgroups taking part (the others exit):
//   row_per_lane   lane l owns row l & 31, half (l >> 5) * 16 B          32 rows x 32 B per instruction (register-only MFMA tail)
//   quad_per_row   8 lanes cover 128 contiguous bytes of a row            8 rows x 128 B per instruction (after a 4 x 4 lane transpose)
//   contiguous     64 lanes x 16 B = 1 KB of one row (2 instr per row)    what an LDS-transposed tail can do
// Build: hipcc --offload-arch=gfx950 -O3 tools/store_pattern_probe.hip -o tools/bin/spp ; run: tools/bin/spp
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int PAT, bool LOAD>
__global__ __launch_bounds__(512) void k(uint16_t* out, const uint16_t* in, int active, int reps, unsigned long long* stamp) {
    if ((int)blockIdx.x >= active) return;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wm = w >> 2, wn = w & 3;
    const size_t ld = 1024;  // elements per row
    unsigned long long t0 = wall_clock64();
    u32x4 acc = u32x4{(unsigned)t, 1u, 2u, 3u};
    for (int r = 0; r < reps; r++) {
        // tile (blockIdx.x * reps + r): rows [tile * 256, +256) x cols [0, 256) of a [M, 1024] matrix
        const size_t row0 = ((size_t)blockIdx.x * reps + r) * 256;
#pragma unroll
        for (int i = 0; i < 4; i++) {
#pragma unroll
            for (int X = 0; X < 4; X++) {
                size_t row, col;
                if (PAT == 0) {  // wave tile 128 x 64: row = wm*128 + i*32 + (lane & 31), col = wn*64 + X*16 + (lane >> 5)*8
                    row = wm * 128 + i * 32 + (lane & 31);
                    col = wn * 64 + X * 16 + (lane >> 5) * 8;
                } else if (PAT == 1) {  // lanes 4g+x (+32h): row = wm*128 + i*32 + 4g + X, col = wn*64 + x*16 + h*8
                    const int g = (lane & 31) >> 2, x = lane & 3, h = lane >> 5;
                    row = wm * 128 + i * 32 + 4 * g + X;
                    col = wn * 64 + x * 16 + h * 8;
                } else {  // wave w owns rows w*32 .. +32 entirely: instruction (i, X) covers row w*32 + i*8 + X*2 + (lane >> 5), col (lane & 31) * 8
                    row = w * 32 + i * 8 + X * 2 + (lane >> 5);
                    col = (lane & 31) * 8;
                }
                uint16_t* p = out + (row0 + row) * ld + col;
                if (LOAD) {
                    const u32x4 v = *reinterpret_cast<const u32x4*>(in + (row0 + row) * ld + col);
                    acc = u32x4{acc.x + v.x, acc.y ^ v.y, acc.z + v.z, acc.w ^ v.w};
                }
                *reinterpret_cast<u32x4*>(p) = acc;
            }
        }
    }
    unsigned long long t1 = wall_clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long t2 = wall_clock64();
    if (lane == 0) {
        stamp[(blockIdx.x * 8 + w) * 3 + 0] = t0;
        stamp[(blockIdx.x * 8 + w) * 3 + 1] = t1;
        stamp[(blockIdx.x * 8 + w) * 3 + 2] = t2;
    }
}

int main() {
    const int reps = 4, cus = 256;
    size_t elems = (size_t)cus * reps * 256 * 1024;
    uint16_t *out, *in;
    unsigned long long* st;
    hipMalloc(&out, elems * 2);
    hipMalloc(&in, elems * 2);
    hipMemset(in, 1, elems * 2);
    hipMalloc(&st, cus * 8 * 3 * 8);
    std::vector<unsigned long long> h(cus * 8 * 3);
    const char* names[3] = {"row_per_lane", "quad_per_row", "contiguous"};
    for (int load = 0; load < 2; load++)
        for (int pat = 0; pat < 3; pat++)
            for (int active : {256, 64, 8}) {
                auto launch = [&] {
                    if (load) {
                        if (pat == 0) hipLaunchKernelGGL((k<0, true>), dim3(cus), dim3(512), 0, 0, out, in, active, reps, st);
                        if (pat == 1) hipLaunchKernelGGL((k<1, true>), dim3(cus), dim3(512), 0, 0, out, in, active, reps, st);
                        if (pat == 2) hipLaunchKernelGGL((k<2, true>), dim3(cus), dim3(512), 0, 0, out, in, active, reps, st);
                    } else {
                        if (pat == 0) hipLaunchKernelGGL((k<0, false>), dim3(cus), dim3(512), 0, 0, out, in, active, reps, st);
                        if (pat == 1) hipLaunchKernelGGL((k<1, false>), dim3(cus), dim3(512), 0, 0, out, in, active, reps, st);
                        if (pat == 2) hipLaunchKernelGGL((k<2, false>), dim3(cus), dim3(512), 0, 0, out, in, active, reps, st);
                    }
                };
                launch();
                launch();
                hipDeviceSynchronize();
                hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
                std::vector<double> issue, drain;
                for (int b = 0; b < active; b++) {
                    unsigned long long a0 = ~0ull, a1 = 0, a2 = 0;
                    for (int w = 0; w < 8; w++) {
                        a0 = std::min(a0, h[(b * 8 + w) * 3]);
                        a1 = std::max(a1, h[(b * 8 + w) * 3 + 1]);
                        a2 = std::max(a2, h[(b * 8 + w) * 3 + 2]);
                    }
                    issue.push_back((a1 - a0) / 100.0 / reps);
                    drain.push_back((a2 - a0) / 100.0 / reps);
                }
                std::sort(issue.begin(), issue.end());
                std::sort(drain.begin(), drain.end());
                printf("%-13s %s active=%3d  per 128 KB tile: issue p50 %.2f us  issue+drain p50 %.2f us p90 %.2f us  (%.1f B/clk/CU at 2.0 GHz)\n",
                       names[pat], load ? "load+store" : "store     ", active, issue[issue.size() / 2], drain[drain.size() / 2],
                       drain[drain.size() * 9 / 10], (load ? 2 : 1) * 131072.0 / (drain[drain.size() / 2] * 2000.0));
            }
    return 0;
}
