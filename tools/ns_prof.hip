// Phase profile of bidiag_reflectors_reg_kernel (in-kernel cycle stamps kept in LDS, waves 0..3, every 8th step).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -DBASQ_NS_PROF tools/ns_prof.hip -o /tmp/ns_prof
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
    {   // bidiag_reflectors_reg_kernel: cycles per phase, summed over the steps, per wave ([wave][slot])
        const char* name[10] = {"barrier 3 -> loop top", "A: loads .. first wave_sum4", "A: other rows", "A: partial-row store", "wait at barrier 1",
                                "B1: norms, H parameters", "B1: 16 partial rows, w, row t+1, norm share", "wait at barrier 2",
                                "B2: G parameters, v", "(unused)"};
        printf("%-36s", "cycles per step (mean over steps)");
        for (int w = 0; w < 16; w += (w < 3 ? 1 : 4)) printf("   wave %2d", w);
        printf("\n");
        for (int sl = 0; sl < 10; ++sl) {
            printf("%-36s", name[sl]);
            for (int w = 0; w < 16; w += (w < 3 ? 1 : 4)) printf(" %9.0f", prof[(size_t)w * 10 + sl] / (double)(m - 1));
            printf("\n");
        }
        double tot = 0;
        for (int sl = 0; sl < 10; ++sl) tot += prof[sl];
        printf("wave 0 total %.0f cycles per step\n", tot / (m - 1));
    }
    return 0;
}
