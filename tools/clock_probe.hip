// Does the shader clock differ between a chip-filling fp64 launch that follows another one and the same launch after an
// idle gap?  (Question behind profiles/r04_h_block_sums_after_idle_or_chain.txt: the round-1 block sums take 6.35 ms back
// to back and 7.5 ms inside a batch.)
//   hipcc --offload-arch=gfx950 -O2 -o tools/clock_probe tools/clock_probe.hip && tools/clock_probe
// Every work-group runs a fixed chain of v_mfma_f64_16x16x4 + fp64 FMAs (the block sums' instruction mix) and records
// s_memrealtime (constant 100 MHz) and s_memtime (shader-clock cycles) before and after: cycles / realtime = the clock the
// work-group ran at.  The launch has ~10 rounds of work-groups, so the profile over the launch's duration is visible.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <unistd.h>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef double d4 __attribute__((ext_vector_type(4)));

struct Rec { unsigned long long rt0, rt1, cy0, cy1; };

// 150 VGPR-ish footprint is not needed here: occupancy is set with LDS (52 KB -> 3 work-groups per CU, as the block sums)
__global__ void __launch_bounds__(256) fp64_load(Rec* rec, double* sink, int trips) {
    __shared__ double pad[6656];
    pad[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const unsigned long long rt0 = wall_clock64(), cy0 = clock64();
    d4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0};
    double a = pad[threadIdx.x & 63] * 1e-3, b = 1.0 + 1e-9 * threadIdx.x, f0 = a, f1 = b, f2 = a + b, f3 = a - b;
    for (int t = 0; t < trips; ++t) {
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, b, acc3, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            f0 = __builtin_fma(f0, b, a);
            f1 = __builtin_fma(f1, b, a);
            f2 = __builtin_fma(f2, b, a);
            f3 = __builtin_fma(f3, b, a);
        }
    }
    const unsigned long long cy1 = clock64(), rt1 = wall_clock64();
    double s = f0 + f1 + f2 + f3;
    for (int r = 0; r < 4; ++r) s += acc0[r] + acc1[r] + acc2[r] + acc3[r];
    if (s == 12345.678) sink[0] = s;
    if (threadIdx.x == 0) rec[blockIdx.x] = Rec{rt0, rt1, cy0, cy1};
}

// Something to keep the chip "busy" beside a single-work-group chain: nwg work-groups of `threads` threads that either sleep
// (s_sleep: occupancy without arithmetic) or run fp64 FMAs until `ticks` 10-ns ticks have passed.
__global__ void heater(double* sink, unsigned long long ticks, int mode) {
    const unsigned long long t0 = wall_clock64();
    double f = threadIdx.x * 1e-3, b = 1.0 + 1e-9 * threadIdx.x;
    while (wall_clock64() - t0 < ticks) {
        if (mode == 0) {
            __builtin_amdgcn_s_sleep(32);
        } else {
#pragma unroll
            for (int u = 0; u < 64; ++u) f = __builtin_fma(f, b, 1e-3);
        }
    }
    if (f == 12345.678) sink[1] = f;
}

static int profile(const char* name, Rec* d_rec, double* d_sink, int nwg, int trips, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
    CHECK(hipEventRecord(e0, st));
    hipLaunchKernelGGL(fp64_load, dim3(nwg), dim3(256), 0, st, d_rec, d_sink, trips);
    CHECK(hipEventRecord(e1, st));
    CHECK(hipStreamSynchronize(st));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<Rec> h(nwg);
    CHECK(hipMemcpy(h.data(), d_rec, sizeof(Rec) * nwg, hipMemcpyDeviceToHost));
    unsigned long long t_min = ~0ull, t_max = 0;
    for (auto& r : h) { t_min = std::min(t_min, r.rt0); t_max = std::max(t_max, r.rt1); }
    const int NB = 8;
    double cyc[NB] = {0}, rt[NB] = {0};
    const double span = (double)(t_max - t_min);
    double wg_us = 0;
    for (auto& r : h) {
        int bin = (int)(NB * ((0.5 * (r.rt0 + r.rt1) - t_min) / span));
        bin = std::min(std::max(bin, 0), NB - 1);
        cyc[bin] += (double)(r.cy1 - r.cy0);
        rt[bin] += (double)(r.rt1 - r.rt0);
        wg_us += (r.rt1 - r.rt0) / 100.0;
    }
    printf("%-44s %6.3f ms (device span %6.3f ms, work-group %5.1f us)  cycles per 10-ns tick over eighths of the launch:", name, ms,
           span / 1e5, wg_us / nwg);
    for (int b = 0; b < NB; ++b) printf(" %5.2f", rt[b] > 0 ? cyc[b] / rt[b] : 0.0);
    printf("\n");
    return 0;
}

int main() {
    const int nwg = 256 * 3 * 10, trips = 2500;
    Rec* d_rec; double* d_sink;
    CHECK(hipMalloc(&d_rec, sizeof(Rec) * nwg));
    CHECK(hipMalloc(&d_sink, 64));
    hipStream_t st; CHECK(hipStreamCreate(&st));
    hipStream_t side; CHECK(hipStreamCreate(&side));
    hipEvent_t e0, e1, ev_go, ev_done; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventCreate(&ev_go)); CHECK(hipEventCreate(&ev_done));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(fp64_load, dim3(nwg), dim3(256), 0, st, d_rec, d_sink, trips);
    CHECK(hipStreamSynchronize(st));
    for (int rep = 0; rep < 2; ++rep) {
        // back to back: two untimed launches in front
        for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(fp64_load, dim3(nwg), dim3(256), 0, st, d_rec, d_sink, trips);
        if (profile("back to back", d_rec, d_sink, nwg, trips, st, e0, e1)) return 1;
        const int gaps_ms[] = {1, 2, 5, 10, 20, 50, 200};
        for (int gms : gaps_ms) {
            for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(fp64_load, dim3(nwg), dim3(256), 0, st, d_rec, d_sink, trips);
            CHECK(hipStreamSynchronize(st));
            usleep(gms * 1000);
            char name[64];
            snprintf(name, sizeof name, "after %d ms idle", gms);
            if (profile(name, d_rec, d_sink, nwg, trips, st, e0, e1)) return 1;
        }
        // a single work-group busy for ~6 ms (the reduction chain's footprint), then the launch
        for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(fp64_load, dim3(nwg), dim3(256), 0, st, d_rec, d_sink, trips);
        hipLaunchKernelGGL(fp64_load, dim3(1), dim3(256), 0, st, d_rec, d_sink, 64000);
        if (profile("after ~6 ms of ONE busy work-group", d_rec, d_sink, nwg, trips, st, e0, e1)) return 1;
        // the same with a heater on a second stream for those 6 ms
        struct { const char* name; int nwg, threads, mode; } heaters[] = {
            {"  + 248 x 1 sleeping wave", 248, 64, 0}, {"  + 248 x 1 wave of FMAs", 248, 64, 1},
            {"  + 248 x 4 waves of FMAs", 248, 256, 1}, {"  + 496 x 4 waves of FMAs", 496, 256, 1}, {"  + 64 x 4 waves of FMAs", 64, 256, 1}};
        for (auto& hsp : heaters) {
            for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(fp64_load, dim3(nwg), dim3(256), 0, st, d_rec, d_sink, trips);
            CHECK(hipEventRecord(ev_go, st));
            CHECK(hipStreamWaitEvent(side, ev_go, 0));
            hipLaunchKernelGGL(heater, dim3(hsp.nwg), dim3(hsp.threads), 0, side, d_sink, 600000ull, hsp.mode);
            CHECK(hipEventRecord(ev_done, side));
            hipLaunchKernelGGL(fp64_load, dim3(1), dim3(256), 0, st, d_rec, d_sink, 64000);
            CHECK(hipStreamWaitEvent(st, ev_done, 0));
            if (profile(hsp.name, d_rec, d_sink, nwg, trips, st, e0, e1)) return 1;
        }
    }
    return 0;
}
