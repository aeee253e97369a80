// microbench.hip -- per-CU issue rates that bound the blocksum kernel on gfx950:
//   v_fma_f64 (VALU fp64), v_mfma_f64_16x16x4_f64 (matrix fp64), and both together.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench tools/microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0 = fma only (two constant operands), 1 = mfma only, 2 = both interleaved (1 mfma : 20 fma),
                      // 3 = fma only with THREE VGPR operands (what a VALU distance kernel issues: a[k] * b[k] + D)
__global__ void __launch_bounds__(256) rate_kernel(double* out, int iters, double seed, unsigned long long* stamps) {
    // in-kernel clock (MI355X_MICROARCH.md, DVFS item 6): d(s_memtime) / d(s_memrealtime) x 100 MHz
    const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    double a0 = seed + threadIdx.x, a1 = a0 * 0.5, a2 = a0 * 0.25, a3 = a0 * 0.125;
    double a4 = a0 + 1, a5 = a0 + 2, a6 = a0 + 3, a7 = a0 + 4;
    const double m = 0.999999, c = 1e-9;
    d4 D0 = {0, 0, 0, 0}, D1 = {0, 0, 0, 0}, D2 = {0, 0, 0, 0}, D3 = {0, 0, 0, 0};
    const double pa = seed * 1e-3, pb = seed * 2e-3;
    double b0 = a0 * 1e-9 + 0.999999, b1 = b0 + 1e-12, b2 = b0 + 2e-12, b3 = b0 + 3e-12;
    double c0 = seed * 1e-9, c1 = c0 * 2, c2 = c0 * 3, c3 = c0 * 4;
    asm volatile("" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
    for (int i = 0; i < iters; ++i) {
        if (MODE == 3) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                a0 = __builtin_fma(a0, b0, c0); a1 = __builtin_fma(a1, b1, c1); a2 = __builtin_fma(a2, b2, c2); a3 = __builtin_fma(a3, b3, c3);
                a4 = __builtin_fma(a4, b0, c1); a5 = __builtin_fma(a5, b1, c2); a6 = __builtin_fma(a6, b2, c3); a7 = __builtin_fma(a7, b3, c0);
            }
        }
        if (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int u = 0; u < (MODE == 2 ? 10 : 16); ++u) {
                a0 = __builtin_fma(a0, m, c); a1 = __builtin_fma(a1, m, c); a2 = __builtin_fma(a2, m, c); a3 = __builtin_fma(a3, m, c);
                a4 = __builtin_fma(a4, m, c); a5 = __builtin_fma(a5, m, c); a6 = __builtin_fma(a6, m, c); a7 = __builtin_fma(a7, m, c);
            }
        }
        if (MODE == 1 || MODE == 2) {
            D0 = __builtin_amdgcn_mfma_f64_16x16x4f64(pa, pb, D0, 0, 0, 0);
            D1 = __builtin_amdgcn_mfma_f64_16x16x4f64(pa, pb, D1, 0, 0, 0);
            D2 = __builtin_amdgcn_mfma_f64_16x16x4f64(pa, pb, D2, 0, 0, 0);
            D3 = __builtin_amdgcn_mfma_f64_16x16x4f64(pa, pb, D3, 0, 0, 0);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + D0[0] + D1[1] + D2[2] + D3[3];
    if (stamps && threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - clk0;
        stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

template <int MODE>
static void run(const char* name, int waves_per_simd, double fma_per_iter, double mfma_per_iter, int warm_reps) {
    const int blocks = 256 * waves_per_simd;   // 256 CUs x (4 waves = 1 per SIMD) per block
    const int iters = 20000;
    double* out;
    hipMalloc(&out, (size_t)blocks * 256 * 8);
    unsigned long long* stamps;
    hipMalloc(&stamps, (size_t)blocks * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    // >= 2 s of back-to-back launches first: the clock the chip HOLDS under this load, not the boost of a cold start
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.0, (unsigned long long*)nullptr);
    hipDeviceSynchronize();
    for (int rep = 0; rep < warm_reps; ++rep)
        hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, (unsigned long long*)nullptr);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, stamps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)blocks * 4;
    const double fma = waves * iters * fma_per_iter;      // wave-level instructions
    const double mfma = waves * iters * mfma_per_iter;
    const double tf_valu = fma * 64 * 2 / (ms * 1e-3) / 1e12;
    const double tf_mfma = mfma * 2048 / (ms * 1e-3) / 1e12;
    // cycles per wave-instruction per SIMD at 2.4 GHz nominal
    const double simd_cycles = ms * 1e-3 * 2.4e9;
    std::vector<unsigned long long> h((size_t)blocks * 2);
    hipMemcpy(h.data(), stamps, (size_t)blocks * 16, hipMemcpyDeviceToHost);
    std::vector<double> ghz;
    for (int b = 0; b < blocks; ++b)
        if (h[2 * b + 1]) ghz.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 0.1);
    std::sort(ghz.begin(), ghz.end());
    const double clk = ghz.empty() ? 0.0 : ghz[ghz.size() / 2];      // median over work-groups, GHz
    const double real_cycles = ms * 1e-3 * clk * 1e9;
    printf("%-22s waves/SIMD=%d  %8.3f ms  in-kernel clock %.3f GHz | VALU %7.2f TF/s (%.2f nominal / %.2f real cyc/fma/SIMD)  "
           "MFMA %7.2f TF/s (%.1f nominal / %.1f real cyc/mfma/SIMD)\n", name,
           waves_per_simd, ms, clk, tf_valu, fma > 0 ? simd_cycles / (fma / 1024.0) : 0.0,
           fma > 0 ? real_cycles / (fma / 1024.0) : 0.0, tf_mfma,
           mfma > 0 ? simd_cycles / (mfma / 1024.0) : 0.0, mfma > 0 ? real_cycles / (mfma / 1024.0) : 0.0);
    hipFree(stamps);
    hipFree(out);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("device: %s  CUs=%d  clock=%d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    for (int w = 1; w <= 4; w *= 2) {
        const int warm = (w == 4) ? 100 : 0;     // the 4-waves/SIMD rows are measured after ~2 s at load
        run<0>("fma_f64 only", w, 128, 0, warm);
        run<1>("mfma_f64 16x16x4 only", w, 0, 4, warm);
        run<2>("both (80 fma : 4 mfma)", w, 80, 4, warm);
        run<3>("fma_f64, 3 VGPR operands", w, 128, 0, warm);
    }
    return 0;
}
