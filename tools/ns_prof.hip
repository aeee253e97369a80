// Phase profile of bidiag_reflectors_reg_kernel (in-kernel clock64 stamps).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -DBASQ_NS_PROF tools/ns_prof.hip -o /tmp/ns_prof
#include "../basq_amd/csrc/basq_hip.hip"
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
    double *dX, *dV, *dtau, *dP;
    long long* dprof;
    hipMalloc(&dX, X.size() * 8);
    hipMalloc(&dV, X.size() * 8);
    hipMalloc(&dtau, m * 8);
    hipMalloc(&dP, (size_t)(n - m) * n * 8);
    hipMalloc(&dprof, (size_t)m * 8 * 16 * 8);
    hipMemset(dprof, 0, (size_t)m * 8 * 16 * 8);
    hipMemcpy(dX, X.data(), X.size() * 8, hipMemcpyHostToDevice);
    hipMemcpyToSymbol(HIP_SYMBOL(g_ns_prof), &dprof, sizeof(dprof));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        int rc = basq_nullspace_f64(dX, m, n, dV, dtau, dP, nullptr, nullptr, nullptr);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("rc=%d  nullspace total %.1f us\n", rc, ms * 1e3);
    }
    std::vector<long long> prof((size_t)m * 8 * 16);
    hipMemcpy(prof.data(), dprof, prof.size() * 8, hipMemcpyDeviceToHost);
    auto at = [&](int t, int slot, int w) { return prof[((size_t)t * 8 + slot) * 16 + w]; };
    printf("kernel span (wave 0): %lld clocks\n", at(m - 2, 3, 0) - at(0, 0, 0));
    printf("  t   A(w0)  A(max)  bar1(w0)   B(w0)  bar2(w0)   step | B:  -    sum16   wait+w   right\n");
    for (int t = 0; t + 1 < m; t += (m > 20 ? m / 12 : 1)) {
        long long amax = 0;
        for (int w = 0; w < 16; ++w) {
            long long d = at(t, 1, w) - at(t, 0, w);
            if (d > amax) amax = d;
        }
        long long next = (t + 2 < m) ? at(t + 1, 0, 0) : at(t, 3, 0);
        printf("%3d %7lld %7lld %9lld %7lld %9lld %7lld | %6lld %7lld %8lld %6lld\n", t, at(t, 1, 0) - at(t, 0, 0), amax,
               at(t, 2, 0) - at(t, 1, 0), at(t, 3, 0) - at(t, 2, 0), next - at(t, 3, 0), next - at(t, 0, 0),
               at(t, 4, 0) - at(t, 2, 0), at(t, 5, 0) - at(t, 4, 0), at(t, 6, 0) - at(t, 5, 0), at(t, 3, 0) - at(t, 6, 0));
    }
    // chol_inv phases (q = m - 1)
    {
        const int q = m > 1 ? m - 1 : 1;
        std::vector<double> Y((size_t)(4 * q + 7) * q), G((size_t)q * q, 0.0);
        for (auto& v : Y) v = rand() / (double)RAND_MAX - 0.5;
        for (int i = 0; i < q; ++i)
            for (int j = 0; j < q; ++j) {
                double acc = 0;
                for (int r = 0; r < 4 * q + 7; ++r) acc += Y[(size_t)r * q + i] * Y[(size_t)r * q + j];
                G[(size_t)i * q + j] = acc;
            }
        double *dG, *dW;
        int* dinfo;
        hipMalloc(&dG, G.size() * 8);
        hipMalloc(&dW, G.size() * 8);
        hipMalloc(&dinfo, 8);
        for (int rep = 0; rep < 3; ++rep) {
            hipMemcpy(dG, G.data(), G.size() * 8, hipMemcpyHostToDevice);
            hipEventRecord(e0);
            int rc = basq_chol_inv_f64(dG, q, dW, dinfo, 1e-13, nullptr);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("rc=%d  chol_inv q=%d total %.1f us\n", rc, q, ms * 1e3);
        }
        hipMemcpy(prof.data(), dprof, 4 * 16 * 8, hipMemcpyDeviceToHost);
        printf("chol_inv clocks (wave 0): cholesky loop %lld, write-back + recip %lld, inverse (wave 0) %lld, inverse (wave 1) %lld\n",
               at(0, 1, 0) - at(0, 0, 0), at(0, 2, 0) - at(0, 1, 0), at(0, 3, 0) - at(0, 2, 0), at(0, 3, 1) - at(0, 2, 1));
    }
    // elimination phases (PhiT = the null space just computed)
    {
        std::vector<double> mu(n);
        double tot = 0;
        for (auto& v : mu) { v = 0.05 + rand() / (double)RAND_MAX; tot += v; }
        for (auto& v : mu) v /= tot;
        double *dmu, *dw;
        int *dkr, *dkept, *dinfo;
        hipMalloc(&dmu, n * 8);
        hipMalloc(&dw, n * 8);
        hipMalloc(&dkr, n * 4);
        hipMalloc(&dkept, n * 4);
        hipMalloc(&dinfo, 8);
        hipMemcpy(dmu, mu.data(), n * 8, hipMemcpyHostToDevice);
        basq_nullspace_f64(dX, m, n, dV, dtau, dP, nullptr, nullptr, nullptr);
        hipEventRecord(e0);
        int rc = basq_car_eliminate_f64(dP, dmu, n, m, dkr, dkept, dw, dinfo, nullptr, nullptr);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("rc=%d  car_eliminate %d x %d total %.1f us\n", rc, m, n, ms * 1e3);
        hipMemcpy(prof.data(), dprof, prof.size() * 8, hipMemcpyDeviceToHost);
        printf("  k  scan+mu(w0)  update+ratio(w0)  update(max over waves)  barrier(w0)  step\n");
        const int nr = n - m;
        for (int k = 0; k + 1 < nr; k += (nr > 20 ? nr / 10 : 1)) {
            long long umax = 0;
            for (int w = 0; w < 16; ++w) umax = std::max(umax, at(k, 4, w) - at(k, 3, w));
            printf("%3d %12lld %17lld %23lld %12lld %6lld\n", k, at(k, 3, 0) - at(k, 0, 0), at(k, 4, 0) - at(k, 3, 0), umax,
                   at(k, 5, 0) - at(k, 4, 0), at(k + 1, 0, 0) - at(k, 0, 0));
        }
    }
    return 0;
}
