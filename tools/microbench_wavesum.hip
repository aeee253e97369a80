// microbench_wavesum.hip -- latency of a 64-lane fp64 sum inside a dependent chain (what the per-round reductions are made of):
//   the DPP form the kernels use (wave_sum / wave_sum4 of basq_reduction.hip) against a form on v_mfma_f64_4x4x4_4b_f64:
//   with B = ones, D[b][i][j] = sum_k A[b][i][k] adds the four 16-lane rows; fed back as A, a second one adds the four
//   lanes i of each block; two DPP rotations (row_ror 4, 8) add the four blocks.  Result checked against the DPP sum.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench_wavesum tools/microbench_wavesum.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_shift_f64(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, ROW_MASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double readlane_f64(double v, int src) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)b, src);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double wave_sum_dpp(double v) {
    v += dpp_shift_f64<0x111, 0xf>(v);
    v += dpp_shift_f64<0x112, 0xf>(v);
    v += dpp_shift_f64<0x114, 0xf>(v);
    v += dpp_shift_f64<0x118, 0xf>(v);
    v += dpp_shift_f64<0x142, 0xa>(v);
    v += dpp_shift_f64<0x143, 0xc>(v);
    return readlane_f64(v, 63);
}
template <int CTRL>
__device__ __forceinline__ double dpp_rot_f64(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_mov_dpp((int)b, CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double wave_sum_mfma(double v, double ones) {
    double s = __builtin_amdgcn_mfma_f64_4x4x4f64(v, ones, 0.0, 0, 0, 0);     // the four rows of 16 lanes
    s = __builtin_amdgcn_mfma_f64_4x4x4f64(s, ones, 0.0, 0, 0, 0);            // the four lanes i of each block
    s += dpp_rot_f64<0x124>(s);                                               // the four blocks
    s += dpp_rot_f64<0x128>(s);
    return readlane_f64(s, 0);
}
__device__ __forceinline__ void wave_sum4_mfma(double& x0, double& x1, double& x2, double& x3, double ones) {
    double s0 = __builtin_amdgcn_mfma_f64_4x4x4f64(x0, ones, 0.0, 0, 0, 0);
    double s1 = __builtin_amdgcn_mfma_f64_4x4x4f64(x1, ones, 0.0, 0, 0, 0);
    double s2 = __builtin_amdgcn_mfma_f64_4x4x4f64(x2, ones, 0.0, 0, 0, 0);
    double s3 = __builtin_amdgcn_mfma_f64_4x4x4f64(x3, ones, 0.0, 0, 0, 0);
    s0 = __builtin_amdgcn_mfma_f64_4x4x4f64(s0, ones, 0.0, 0, 0, 0);
    s1 = __builtin_amdgcn_mfma_f64_4x4x4f64(s1, ones, 0.0, 0, 0, 0);
    s2 = __builtin_amdgcn_mfma_f64_4x4x4f64(s2, ones, 0.0, 0, 0, 0);
    s3 = __builtin_amdgcn_mfma_f64_4x4x4f64(s3, ones, 0.0, 0, 0, 0);
    s0 += dpp_rot_f64<0x124>(s0); s1 += dpp_rot_f64<0x124>(s1); s2 += dpp_rot_f64<0x124>(s2); s3 += dpp_rot_f64<0x124>(s3);
    s0 += dpp_rot_f64<0x128>(s0); s1 += dpp_rot_f64<0x128>(s1); s2 += dpp_rot_f64<0x128>(s2); s3 += dpp_rot_f64<0x128>(s3);
    x0 = readlane_f64(s0, 0); x1 = readlane_f64(s1, 0); x2 = readlane_f64(s2, 0); x3 = readlane_f64(s3, 0);
}
__device__ __forceinline__ void wave_sum4_dpp(double& x0, double& x1, double& x2, double& x3) {   // as in basq_reduction.hip
    auto fold32 = [](double a, double b) {
        const long long ba = __double_as_longlong(a), bb = __double_as_longlong(b);
        const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)ba, (unsigned)bb, false, false);
        const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(ba >> 32), (unsigned)(bb >> 32), false, false);
        return __longlong_as_double(((long long)hi[0] << 32) | lo[0]) + __longlong_as_double(((long long)hi[1] << 32) | lo[1]);
    };
    auto fold16 = [](double a, double b) {
        const long long ba = __double_as_longlong(a), bb = __double_as_longlong(b);
        const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)ba, (unsigned)bb, false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(ba >> 32), (unsigned)(bb >> 32), false, false);
        return __longlong_as_double(((long long)hi[0] << 32) | lo[0]) + __longlong_as_double(((long long)hi[1] << 32) | lo[1]);
    };
    auto perm = [](double v, auto tag) {
        constexpr int CTRL = decltype(tag)::value;
        const long long b = __double_as_longlong(v);
        const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xf, 0xf, true);
        const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, true);
        return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
    };
    double v = fold16(fold32(x0, x1), fold32(x2, x3));
    v += perm(v, std::integral_constant<int, 0xB1>{});
    v += perm(v, std::integral_constant<int, 0x4E>{});
    v += perm(v, std::integral_constant<int, 0x141>{});
    v += perm(v, std::integral_constant<int, 0x140>{});
    x0 = readlane_f64(v, 0); x2 = readlane_f64(v, 16); x1 = readlane_f64(v, 32); x3 = readlane_f64(v, 48);
}

template <int MODE>   // 0 dpp sum, 1 mfma sum, 2 dpp sum4, 3 mfma sum4
__global__ void chain_kernel(double* out, int iters, long long* cyc) {
    const int lane = threadIdx.x;
    double ones = 1.0;
    asm volatile("" : "+v"(ones));
    double x = 1.0 + lane * 1e-3, y = 0.5 + lane * 2e-3, z = 0.25 + lane * 3e-3, w = 0.125 + lane * 5e-4;
    const double a = 1e-3 * (lane + 1);
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) x = wave_sum_dpp(x) * a + 1e-3;
        if (MODE == 1) x = wave_sum_mfma(x, ones) * a + 1e-3;
        if (MODE == 2) { wave_sum4_dpp(x, y, z, w); x = x * a + 1e-3; y = y * a + 2e-3; z = z * a + 3e-3; w = w * a + 4e-3; }
        if (MODE == 3) { wave_sum4_mfma(x, y, z, w, ones); x = x * a + 1e-3; y = y * a + 2e-3; z = z * a + 3e-3; w = w * a + 4e-3; }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[lane] = x + y + z + w;
    if (lane == 0) cyc[0] = t1 - t0;
}

__global__ void check_kernel(double* out) {
    const int lane = threadIdx.x;
    double ones = 1.0;
    asm volatile("" : "+v"(ones));
    const double v = sin(0.37 * lane) + 1e-3 * lane;
    out[lane] = wave_sum_dpp(v);
    out[64 + lane] = wave_sum_mfma(v, ones);
    double a = v, b = 2 * v, c = v * v, d = 1.0 / (1.0 + lane);
    wave_sum4_mfma(a, b, c, d, ones);
    out[128 + lane] = a; out[192 + lane] = b; out[256 + lane] = c; out[320 + lane] = d;
    double a2 = v, b2 = 2 * v, c2 = v * v, d2 = 1.0 / (1.0 + lane);
    wave_sum4_dpp(a2, b2, c2, d2);
    out[384 + lane] = a2; out[448 + lane] = b2; out[512 + lane] = c2; out[576 + lane] = d2;
}

int main() {
    double* out; long long* cyc;
    hipMalloc(&out, 1024 * 8); hipMalloc(&cyc, 8);
    double h[640];
    hipLaunchKernelGGL(check_kernel, dim3(1), dim3(64), 0, 0, out);
    hipMemcpy(h, out, 640 * 8, hipMemcpyDeviceToHost);
    double ref = 0; for (int l = 0; l < 64; ++l) ref += sin(0.37 * l) + 1e-3 * l;
    printf("sum: host %.17g  dpp %.17g  mfma %.17g (lane 0), mfma lane 37 %.17g\n", ref, h[0], h[64], h[64 + 37]);
    printf("sum4 mfma: %.17g %.17g %.17g %.17g\nsum4 dpp : %.17g %.17g %.17g %.17g\n", h[128], h[192], h[256], h[320], h[384], h[448],
           h[512], h[576]);
    const int iters = 20000;
    const char* names[4] = {"wave_sum  (DPP, 6 stages + readlane)", "wave_sum  (2 MFMA 4x4x4 + 2 DPP rotations + readlane)",
                            "wave_sum4 (permlane swaps + DPP butterfly)", "wave_sum4 (8 MFMA 4x4x4 + 8 DPP rotations)"};
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) hipLaunchKernelGGL(chain_kernel<0>, dim3(1), dim3(64), 0, 0, out, iters, cyc);
            if (mode == 1) hipLaunchKernelGGL(chain_kernel<1>, dim3(1), dim3(64), 0, 0, out, iters, cyc);
            if (mode == 2) hipLaunchKernelGGL(chain_kernel<2>, dim3(1), dim3(64), 0, 0, out, iters, cyc);
            if (mode == 3) hipLaunchKernelGGL(chain_kernel<3>, dim3(1), dim3(64), 0, 0, out, iters, cyc);
            hipDeviceSynchronize();
        }
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-56s %8.1f s_memtime ticks per dependent reduction (+ one fma)\n", names[mode], (double)c / iters);
    }
    return 0;
}
