// Phase profile of the cluster kernels (in-kernel clock64 stamps), one CU form.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -DBASQ_NS_PROF tools/ns_prof2.hip -o /tmp/ns_prof2
#include "../basq_amd/csrc/basq_reduction.hip"
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <vector>

int main(int argc, char** argv) {
    const int m = argc > 1 ? atoi(argv[1]) : 100, n = argc > 2 ? atoi(argv[2]) : 200;
    std::vector<double> X((size_t)m * n);
    srand(1);
    for (auto& v : X) v = rand() / (double)RAND_MAX - 0.5;
    for (int c = 0; c < n; ++c) X[c] = 1.0;
    double *dX, *dV, *dtau, *dP, *dws;
    long long* dprof;
    const size_t nprof = (size_t)(m + n) * 8 * 16;
    hipMalloc(&dX, X.size() * 8);
    hipMalloc(&dV, X.size() * 8);
    hipMalloc(&dtau, m * 8);
    hipMalloc(&dP, (size_t)(n - m) * n * 8);
    hipMalloc(&dprof, nprof * 8);
    const long long wsn = basq_reduction_ws_doubles(m, n);
    hipMalloc(&dws, (wsn > 0 ? wsn : 16) * 8);
    hipMemset(dprof, 0, nprof * 8);
    hipMemcpy(dX, X.data(), X.size() * 8, hipMemcpyHostToDevice);
    hipMemcpyToSymbol(HIP_SYMBOL(g_ns_prof), &dprof, sizeof(dprof));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        int rc = basq_nullspace_f64(dX, m, n, dV, dtau, dP, wsn > 0 ? dws : nullptr, nullptr, nullptr);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("rc=%d  nullspace total %.1f us\n", rc, ms * 1e3);
    }
    std::vector<long long> prof(nprof);
    hipMemcpy(prof.data(), dprof, prof.size() * 8, hipMemcpyDeviceToHost);
    auto at = [&](int t, int slot, int w) { return prof[((size_t)t * 8 + slot) * 16 + w]; };
    printf("bidiag span (wave 0): %lld clocks for %d steps\n", at(m - 1, 3, 0) - at(0, 0, 0), m);
    printf(" step  bulk(w0) bulk(max)  barrier(w0)  sum+reads  H+w+rn  wave_sum  G+v(store)   total\n");
    for (int t = 0; t < m; t += (m > 20 ? m / 12 : 1)) {
        long long amax = 0;
        for (int w = 0; w < 8; ++w) amax = std::max(amax, at(t, 1, w) - at(t, 0, w));
        long long next = (t + 1 < m) ? at(t + 1, 0, 0) : at(t, 3, 0);
        printf("%4d %9lld %9lld %12lld %10lld %7lld %9lld %11lld %7lld\n", t, at(t, 1, 0) - at(t, 0, 0), amax,
               at(t, 2, 0) - at(t, 1, 0), at(t, 4, 0) - at(t, 2, 0), at(t, 5, 0) - at(t, 4, 0), at(t, 6, 0) - at(t, 5, 0),
               at(t, 3, 0) - at(t, 6, 0), next - at(t, 0, 0));
    }
    return 0;
}
