// Phase profile of car_eliminate_ring_kernel (in-kernel clock64 stamps: producer slots 0..4, consumers 5..6).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -DBASQ_NS_PROF tools/car_prof.hip -o tools/car_prof
#include "../basq_amd/csrc/basq_reduction.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char** argv) {
    const int s = argc > 1 ? atoi(argv[1]) : 100, M = argc > 2 ? atoi(argv[2]) : 200;
    const int nrows = M - s, NR = nrows <= 64 ? 4 : 7;
    std::vector<double> P((size_t)nrows * M), mu(M);
    srand(1);
    for (auto& v : P) v = rand() / (double)RAND_MAX - 0.5;
    for (auto& v : mu) v = (0.05 + rand() / (double)RAND_MAX) / M;
    double *dP, *dmu, *dw;
    int *dkr, *dkept, *dinfo;
    long long* dprof;
    const size_t nprof = (size_t)nrows * 8 * 16;
    hipMalloc(&dP, P.size() * 8);
    hipMalloc(&dmu, M * 8);
    hipMalloc(&dw, M * 8);
    hipMalloc(&dkr, M * 4);
    hipMalloc(&dkept, M * 4);
    hipMalloc(&dinfo, 8);
    hipMalloc(&dprof, nprof * 8);
    hipMemset(dprof, 0, nprof * 8);
    hipMemcpy(dP, P.data(), P.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dmu, mu.data(), M * 8, hipMemcpyHostToDevice);
    hipMemcpyToSymbol(HIP_SYMBOL(g_ns_prof), &dprof, sizeof(dprof));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        int rc = basq_car_eliminate_f64(dP, dmu, M, s, dkr, dkept, dw, dinfo, nullptr, nullptr);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        int info[2];
        hipMemcpy(info, dinfo, 8, hipMemcpyDeviceToHost);
        printf("rc=%d  car_eliminate total %.1f us  (kept %d, status %d)\n", rc, ms * 1e3, info[0], info[1]);
    }
    std::vector<long long> prof(nprof);
    hipMemcpy(prof.data(), dprof, nprof * 8, hipMemcpyDeviceToHost);
    auto at = [&](int t, int slot, int w) { return prof[((size_t)t * 8 + slot) * 16 + w]; };
    printf("span: %lld clocks for %d pivots\n", at(nrows - 1, 4, (nrows - 1) / NR) - at(0, 0, 0), nrows);
    printf("  k own | test  argmin  publish  apply | step (to the next test) | next producer: sees it after, applied after | slowest consumer applied after\n");
    for (int k = 0; k + 1 < nrows; ++k) {
        const int w = k / NR, wn = (k + 1) / NR;
        const long long t0 = at(k, 0, w);
        long long next_seen = 0, next_done = 0, slow = 0;
        if (w + 1 <= (nrows - 1) / NR) {
            next_seen = at(k, 5, w + 1) - at(k, 3, w);
            next_done = at(k, 6, w + 1) - at(k, 3, w);
        }
        for (int c = w + 1; c <= (nrows - 1) / NR; ++c) {
            const long long d = at(k, 6, c) - at(k, 3, w);
            if (d > slow) slow = d;
        }
        if (k % (nrows > 24 ? 3 : 1) == 0 || k % NR == NR - 1)
            printf("%3d %3d | %5lld %6lld %7lld %6lld | %6lld | %7lld %7lld | %7lld\n", k, w, at(k, 1, w) - t0, at(k, 2, w) - at(k, 1, w),
                   at(k, 3, w) - at(k, 2, w), at(k, 4, w) - at(k, 3, w), at(k + 1, 0, wn) - t0, next_seen, next_done, slow);
    }
    return 0;
}
