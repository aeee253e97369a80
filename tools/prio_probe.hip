// Does a HIGH-PRIORITY stream get a whole CU for a one-work-group kernel while a chip-filling grid of another stream keeps
// refilling every CU?  (The reductions of a batch need a whole CU -- 1024 threads x 128 VGPRs, or 160 KB of LDS -- and never
// started beside another batch's block sums: DESIGN.md 8.11.)
//   hipcc --offload-arch=gfx950 -O2 -o tools/prio_probe tools/prio_probe.hip && tools/prio_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <unistd.h>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// the block sums' footprint: 256 threads, 3 work-groups per CU (52 KB of LDS each), ~wg_ticks x 10 ns per work-group
__global__ void __launch_bounds__(256) fill(unsigned long long* first_start, unsigned long long* last_end, unsigned long long wg_ticks) {
    __shared__ double pad[6656];
    pad[threadIdx.x] = threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) atomicMin(first_start, t0);
    double f = pad[threadIdx.x & 63];
    while (wall_clock64() - t0 < wg_ticks) {
#pragma unroll
        for (int u = 0; u < 32; ++u) f = __builtin_fma(f, 1.0000001, 1e-9);
    }
    if (f == 1234.5) pad[0] = f;
    if (threadIdx.x == 0) atomicMax(last_end, wall_clock64());
}

// the reductions' footprint: ONE work-group of 1024 threads with all of a CU's LDS
__global__ void __launch_bounds__(1024) whole_cu(unsigned long long* start_end, unsigned long long ticks) {
    extern __shared__ double big[];
    big[threadIdx.x] = threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) { }
    if (threadIdx.x == 0) { start_end[0] = t0; start_end[1] = wall_clock64() + (unsigned long long)(big[1] * 0.0); }
}
// the same as 8 waves with a modest register / LDS footprint (half a CU)
__global__ void __launch_bounds__(512) half_cu(unsigned long long* start_end, unsigned long long ticks) {
    __shared__ double sm[2048];
    sm[threadIdx.x] = threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) { }
    if (threadIdx.x == 0) { start_end[0] = t0; start_end[1] = wall_clock64() + (unsigned long long)(sm[1] * 0.0); }
}

int main() {
    int lo = 0, hi = 0;
    CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    printf("stream priority range: least %d .. greatest %d\n", lo, hi);
    hipStream_t sa, sb_norm, sb_high;
    CHECK(hipStreamCreateWithPriority(&sa, hipStreamNonBlocking, lo));
    CHECK(hipStreamCreateWithPriority(&sb_norm, hipStreamNonBlocking, lo));
    CHECK(hipStreamCreateWithPriority(&sb_high, hipStreamNonBlocking, hi));
    unsigned long long* d;
    CHECK(hipMalloc(&d, 64));
    CHECK(hipFuncSetAttribute((const void*)whole_cu, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int rounds = 24;
    const unsigned long long wg_ticks = 10000;            // 100 us per work-group -> 2.4 ms per fill launch
    for (int rep = 0; rep < 2; ++rep)
        for (int which = 0; which < 4; ++which) {
            const bool high = which & 1, whole = which < 2;
            unsigned long long h[4] = {~0ull, 0ull, 0ull, 0ull};
            CHECK(hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(fill, dim3(256 * 3 * rounds), dim3(256), 0, sa, d, d + 1, wg_ticks);
            usleep(300);                                  // the fill grid is resident
            hipStream_t sb = high ? sb_high : sb_norm;
            if (whole) hipLaunchKernelGGL(whole_cu, dim3(1), dim3(1024), 160 * 1024, sb, d + 2, 20000ull);
            else hipLaunchKernelGGL(half_cu, dim3(1), dim3(512), 0, sb, d + 2, 20000ull);
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
            printf("%-9s work-group on a %-6s-priority stream: starts %8.1f us after the fill grid's first work-group (fill grid runs %8.1f us)\n",
                   whole ? "whole-CU" : "half-CU", high ? "HIGH" : "normal", (double)(h[2] - h[0]) / 100.0, (double)(h[1] - h[0]) / 100.0);
        }
    return 0;
}
