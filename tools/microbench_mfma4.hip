// microbench_mfma4.hip -- is v_mfma_f64_4x4x4_4b_f64 a faster fp64 matrix path than v_mfma_f64_16x16x4_f64 on gfx950?
//   (1) operand/result lane map of the 4x4x4 (4 blocks) form, found by one-hot probing;
//   (2) issue rate alone, and interleaved with v_fma_f64 (does it share the fp64 pipe like the 16x16x4 form does?).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench_mfma4 tools/microbench_mfma4.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void probe_kernel(int* out) {   // one wave; out[la * 64 + lb] = D lane that receives A(la) * B(lb), or -1
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
            const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            const unsigned long long m = __ballot(d != 0.0);
            if (lane == 0) out[la * 64 + lb] = m ? (int)__builtin_ctzll(m) + 64 * (__builtin_popcountll(m) - 1) : -1;
        }
}

// pseudo-random operand in [1, 2): full mantissa toggling between consecutive instructions (power, not just issue rate)
__device__ __forceinline__ double rnd_operand(unsigned long long x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return __longlong_as_double((long long)((x >> 12) | 0x3ff0000000000000ULL)) - 1.5;
}

template <int MODE>   // 0 = 4x4x4 only, 1 = 16x16x4 only, 2 = 4x4x4 + fma (16 mfma : 80 fma), 3 = 16x16x4 + fma (4 : 80),
                      // 4 / 5 = modes 0 / 1 with sixteen different random operand pairs (data-dependent power)
__global__ void __launch_bounds__(256) rate_kernel(double* out, int iters, double seed, unsigned long long* stamps) {
    const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    double a0 = seed + threadIdx.x, a1 = a0 * 0.5, a2 = a0 * 0.25, a3 = a0 * 0.125;
    double a4 = a0 + 1, a5 = a0 + 2, a6 = a0 + 3, a7 = a0 + 4;
    const double m = 0.999999, c = 1e-9;
    double pa = seed * 1e-3 + threadIdx.x * 1e-6, pb = seed * 2e-3 - threadIdx.x * 1e-6;
    asm volatile("" : "+v"(pa), "+v"(pb));
    double e[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) e[u] = 0.0;
    d4 D0 = {0, 0, 0, 0}, D1 = {0, 0, 0, 0}, D2 = {0, 0, 0, 0}, D3 = {0, 0, 0, 0};
    double ra[16], rb[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        ra[u] = rnd_operand(threadIdx.x * 131 + u * 7919 + 1);
        rb[u] = rnd_operand(threadIdx.x * 977 + u * 104729 + 5);
        asm volatile("" : "+v"(ra[u]), "+v"(rb[u]));
    }
    for (int i = 0; i < iters; ++i) {
        if (MODE == 4) {
#pragma unroll
            for (int u = 0; u < 16; ++u) e[u] = __builtin_amdgcn_mfma_f64_4x4x4f64(ra[u], rb[(u + 5) & 15], e[u], 0, 0, 0);
        }
        if (MODE == 5) {
            D0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ra[0], rb[1], D0, 0, 0, 0);
            D1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ra[2], rb[3], D1, 0, 0, 0);
            D2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ra[4], rb[5], D2, 0, 0, 0);
            D3 = __builtin_amdgcn_mfma_f64_16x16x4f64(ra[6], rb[7], D3, 0, 0, 0);
        }
        if (MODE >= 2) {
#pragma unroll
            for (int u = 0; u < 10; ++u) {
                a0 = __builtin_fma(a0, m, c); a1 = __builtin_fma(a1, m, c); a2 = __builtin_fma(a2, m, c); a3 = __builtin_fma(a3, m, c);
                a4 = __builtin_fma(a4, m, c); a5 = __builtin_fma(a5, m, c); a6 = __builtin_fma(a6, m, c); a7 = __builtin_fma(a7, m, c);
            }
        }
        if (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int u = 0; u < 16; ++u) e[u] = __builtin_amdgcn_mfma_f64_4x4x4f64(pa, pb, e[u], 0, 0, 0);
        }
        if (MODE == 1 || MODE == 3) {
            D0 = __builtin_amdgcn_mfma_f64_16x16x4f64(pa, pb, D0, 0, 0, 0);
            D1 = __builtin_amdgcn_mfma_f64_16x16x4f64(pa, pb, D1, 0, 0, 0);
            D2 = __builtin_amdgcn_mfma_f64_16x16x4f64(pa, pb, D2, 0, 0, 0);
            D3 = __builtin_amdgcn_mfma_f64_16x16x4f64(pa, pb, D3, 0, 0, 0);
        }
    }
    double s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + D0[0] + D1[1] + D2[2] + D3[3];
#pragma unroll
    for (int u = 0; u < 16; ++u) s += e[u];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (stamps && threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - clk0;
        stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

template <int MODE>
static void run(const char* name, int waves_per_simd, double fma_per_iter, double mfma_flop_per_iter, int warm_reps) {
    const int blocks = 256 * waves_per_simd;
    const int iters = 20000;
    double* out;
    hipMalloc(&out, (size_t)blocks * 256 * 8);
    unsigned long long* stamps;
    hipMalloc(&stamps, (size_t)blocks * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
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
    const double tf_valu = waves * iters * fma_per_iter * 128 / (ms * 1e-3) / 1e12;
    const double tf_mfma = waves * iters * mfma_flop_per_iter / (ms * 1e-3) / 1e12;
    std::vector<unsigned long long> h((size_t)blocks * 2);
    hipMemcpy(h.data(), stamps, (size_t)blocks * 16, hipMemcpyDeviceToHost);
    std::vector<double> ghz;
    for (int b = 0; b < blocks; ++b)
        if (h[2 * b + 1]) ghz.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 0.1);
    std::sort(ghz.begin(), ghz.end());
    const double clk = ghz.empty() ? 0.0 : ghz[ghz.size() / 2];
    // real cycles one SIMD spends per iteration of its (waves_per_simd) waves
    const double cyc_iter = ms * 1e-3 * clk * 1e9 / iters / waves_per_simd;
    printf("%-26s waves/SIMD=%d  %8.3f ms  clock %.3f GHz | VALU %6.2f TF/s  MFMA %6.2f TF/s | %.1f cycles per wave-iteration\n",
           name, waves_per_simd, ms, clk, tf_valu, tf_mfma, cyc_iter);
    hipFree(stamps);
    hipFree(out);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("device: %s  CUs=%d\n", p.name, p.multiProcessorCount);
    int* tab; hipMalloc(&tab, 4096 * 4);
    hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, 0, tab);
    std::vector<int> h(4096);
    hipMemcpy(h.data(), tab, 4096 * 4, hipMemcpyDeviceToHost);
    printf("lane map of v_mfma_f64_4x4x4_4b: rows = A lane, entries = 'B lane -> D lane' for the non-zero products\n");
    for (int la = 0; la < 64; ++la) {
        printf("A%02d:", la);
        for (int lb = 0; lb < 64; ++lb) if (h[la * 64 + lb] >= 0) printf(" B%02d->D%02d%s", lb, h[la * 64 + lb] % 64, h[la * 64 + lb] >= 64 ? "(+)" : "");
        printf("\n");
    }
    for (int w = 1; w <= 4; w *= 2) {
        const int warm = (w == 4) ? 60 : 0;
        run<0>("mfma 4x4x4_4b only", w, 0, 16 * 512.0, warm);
        run<1>("mfma 16x16x4 only", w, 0, 4 * 2048.0, warm);
        run<2>("4x4x4_4b + fma (16:80)", w, 80, 16 * 512.0, warm);
        run<3>("16x16x4 + fma (4:80)", w, 80, 4 * 2048.0, warm);
        run<4>("4x4x4_4b, random operands", w, 0, 16 * 512.0, warm);
        run<5>("16x16x4, random operands", w, 0, 4 * 2048.0, warm);
    }
    return 0;
}
