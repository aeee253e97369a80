// How many Newton steps do v_rsq_f64 / v_rcp_f64 need?  Max error in ulps of sqrt(d), 1/sqrt(d), 1/d after 1, 2, 3 coupled steps
// against the host's correctly rounded results, over 4M random doubles in [1e-6, 1e6].
//   hipcc --offload-arch=gfx950 -O3 tools/newton_probe.hip -o tools/newton_probe && tools/newton_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
template <int IT>
__global__ void k(const double* x, double* root, double* rroot, double* rcp, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double d = x[i];
    const double y = __builtin_amdgcn_rsq(d);
    double g = d * y, h = 0.5 * y;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const double r = __builtin_fma(-h, g, 0.5);
        g = __builtin_fma(g, r, g);
        h = __builtin_fma(h, r, h);
    }
    root[i] = g;
    rroot[i] = 2.0 * h;
    double z = __builtin_amdgcn_rcp(d);
#pragma unroll
    for (int it = 0; it < IT; ++it) z = __builtin_fma(__builtin_fma(-d, z, 1.0), z, z);
    rcp[i] = z;
}
static double ulps(double got, long double want) {
    int e;
    frexp((double)want, &e);
    return (double)fabsl((long double)got - want) / ldexp(1.0, e - 53);
}
int main() {
    const int n = 1 << 22;
    std::vector<double> x(n), a(n), b(n), c(n);
    srand(7);
    for (auto& v : x) v = exp((rand() / (double)RAND_MAX * 2 - 1) * 13.8) * (1.0 + rand() / (double)RAND_MAX);
    double *dx, *da, *db, *dc;
    hipMalloc(&dx, n * 8); hipMalloc(&da, n * 8); hipMalloc(&db, n * 8); hipMalloc(&dc, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    for (int it = 0; it <= 3; ++it) {
        if (it == 0) k<0><<<n / 256, 256>>>(dx, da, db, dc, n);
        if (it == 1) k<1><<<n / 256, 256>>>(dx, da, db, dc, n);
        if (it == 2) k<2><<<n / 256, 256>>>(dx, da, db, dc, n);
        if (it == 3) k<3><<<n / 256, 256>>>(dx, da, db, dc, n);
        hipMemcpy(a.data(), da, n * 8, hipMemcpyDeviceToHost);
        hipMemcpy(b.data(), db, n * 8, hipMemcpyDeviceToHost);
        hipMemcpy(c.data(), dc, n * 8, hipMemcpyDeviceToHost);
        double e1 = 0, e2 = 0, e3 = 0;
        for (int i = 0; i < n; ++i) {
            e1 = fmax(e1, ulps(a[i], sqrtl((long double)x[i])));
            e2 = fmax(e2, ulps(b[i], 1.0L / sqrtl((long double)x[i])));
            e3 = fmax(e3, ulps(c[i], 1.0L / (long double)x[i]));
        }
        printf("%d Newton steps: max error sqrt %.3g ulp, 1/sqrt %.3g ulp, 1/d %.3g ulp\n", it, e1, e2, e3);
    }
    return 0;
}
