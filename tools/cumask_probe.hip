// Which physical CUs does a CU-masked HIP stream run on?  (hipExtStreamCreateWithCUMask on MI355X: 8 XCDs x 32 CUs)
//   hipcc --offload-arch=gfx950 -O2 -o tools/cumask_probe tools/cumask_probe.hip && tools/cumask_probe
// Every work-group records HW_REG_XCC_ID and HW_REG_HW_ID; the host prints the set of (xcc, se, sh, cu) per mask, and times
// a spin kernel on the full chip vs on the complement of a reserved set.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <set>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void where_am_i(uint32_t* out, int spin) {
    uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);      // HW_REG_XCC_ID[3:0]
    uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);       // HW_REG_HW_ID
    long long t0 = clock64();
    while (clock64() - t0 < spin) { }
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
}

// The stand-in for the reductions: one 1024-thread work-group that needs ALL of a CU's LDS (like car_eliminate_lds_kernel).
__global__ void __launch_bounds__(1024) whole_cu(uint32_t* out, int spin) {
    extern __shared__ double big[];
    uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
    uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    big[threadIdx.x] = (double)hw;
    long long t0 = clock64();
    while (clock64() - t0 < spin) { }
    if (threadIdx.x == 0) { out[0] = xcc; out[1] = hw + (uint32_t)(big[1] * 0.0); }
}

// A chip-filling PERSISTENT grid (3 work-groups of 256 threads per CU, limited by 52 KB of LDS each, like the block sums'
// register-limited occupancy) whose work-groups retire at once when they find themselves on CU 0 / SE 0 / SH 0 of their XCD:
// "software CU reservation" without a queue mask.
__global__ void __launch_bounds__(256) persistent_spin(uint32_t* out, long long spin, int reserve) {
    __shared__ double pad[6656];                                               // 52 KB -> 3 work-groups per CU
    uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    const bool on_reserved = ((hw >> 8) & 0xf) == 0 && ((hw >> 12) & 1) == 0 && ((hw >> 13) & 7) == 0;
    if (reserve && on_reserved) return;
    pad[threadIdx.x] = (double)hw;
    long long t0 = clock64();
    while (clock64() - t0 < spin) { }
    if (threadIdx.x == 0) out[blockIdx.x] = (uint32_t)pad[0];
}

static int run(hipStream_t st, const char* name, uint32_t* d_out, int nwg) {
    std::vector<uint32_t> h(2 * nwg);
    hipLaunchKernelGGL(where_am_i, dim3(nwg), dim3(1024), 0, st, d_out, 20000);
    CHECK(hipStreamSynchronize(st));
    CHECK(hipMemcpy(h.data(), d_out, h.size() * 4, hipMemcpyDeviceToHost));
    std::set<uint32_t> cus;
    int per_xcc[16] = {0};
    for (int i = 0; i < nwg; ++i) {
        uint32_t xcc = h[2 * i] & 0xf, hw = h[2 * i + 1];
        uint32_t cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        uint32_t key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
        if (cus.insert(key).second) per_xcc[xcc]++;
    }
    printf("%-34s: %3zu distinct CUs; per XCC:", name, cus.size());
    for (int x = 0; x < 8; ++x) printf(" %d", per_xcc[x]);
    if (cus.size() <= 16) { printf("  ="); for (uint32_t k : cus) printf(" x%u.se%u.sh%u.cu%u", k >> 12, (k >> 8) & 7, (k >> 4) & 1, k & 0xf); }
    printf("\n");
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s, %d CUs\n", prop.name, prop.multiProcessorCount);
    const int nwg = 4096;
    uint32_t* d_out;
    CHECK(hipMalloc(&d_out, 2 * nwg * 4));
    hipStream_t plain;
    CHECK(hipStreamCreate(&plain));
    if (run(plain, "plain stream", d_out, nwg)) return 1;
    const int words = 8;                                        // 256 bits
    struct { const char* name; uint32_t m[8]; } masks[] = {
        {"all but bits 0..7", {0xffffff00u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u}},
        {"only bits 0..7", {0x000000ffu, 0, 0, 0, 0, 0, 0, 0}},
        {"only bits 0..31", {~0u, 0, 0, 0, 0, 0, 0, 0}},
        {"only bits 0,8,16,..,56", {0x01010101u, 0x01010101u, 0, 0, 0, 0, 0, 0}},
        {"only bit 0", {1u, 0, 0, 0, 0, 0, 0, 0}},
        {"only bit 1", {2u, 0, 0, 0, 0, 0, 0, 0}},
        {"only bit 8", {0x100u, 0, 0, 0, 0, 0, 0, 0}},
        {"only bits 248..255", {0, 0, 0, 0, 0, 0, 0, 0xff000000u}},
    };
    for (auto& mk : masks) {
        hipStream_t st;
        hipError_t e = hipExtStreamCreateWithCUMask(&st, words, mk.m);
        if (e != hipSuccess) { printf("%-34s: hipExtStreamCreateWithCUMask failed: %s\n", mk.name, hipGetErrorString(e)); continue; }
        if (run(st, mk.name, d_out, nwg)) return 1;
        CHECK(hipStreamDestroy(st));
    }
    // concurrency: a long full-occupancy kernel on the masked "wide" stream, then a 1-work-group kernel on the reserved CUs
    hipStream_t wide, chain;
    uint32_t mw[8] = {0xffffff00u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u}, mc[8] = {0xffu, 0, 0, 0, 0, 0, 0, 0};
    CHECK(hipExtStreamCreateWithCUMask(&wide, words, mw));
    CHECK(hipExtStreamCreateWithCUMask(&chain, words, mc));
    hipEvent_t a, b, c, d;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b)); CHECK(hipEventCreate(&c)); CHECK(hipEventCreate(&d));
    uint32_t* d_out2;
    CHECK(hipMalloc(&d_out2, 2 * 65536 * 4));
    hipStream_t allmask, hiprio;
    uint32_t ma[8] = {~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u};
    CHECK(hipExtStreamCreateWithCUMask(&allmask, words, ma));
    int lo_p, hi_p;
    CHECK(hipDeviceGetStreamPriorityRange(&lo_p, &hi_p));
    CHECK(hipStreamCreateWithPriority(&hiprio, hipStreamNonBlocking, hi_p));
    printf("stream priority range: least %d .. greatest %d\n", lo_p, hi_p);
    const char* names[] = {"plain streams", "CU-masked streams (wide: all but 0..7, chain: 0..7)",
                           "wide: masked with ALL bits set, chain: plain", "wide: plain, chain: high-priority stream",
                           "wide: plain, chain: masked 0..7"};
    for (int variant = 0; variant < 5; ++variant) {
        hipStream_t ch2;
        CHECK(hipStreamCreate(&ch2));
        hipStream_t w = plain, ch = ch2;
        if (variant == 1) { w = wide; ch = chain; }
        if (variant == 2) { w = allmask; }
        if (variant == 3) { ch = hiprio; }
        if (variant == 4) { ch = chain; }
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(a, w));
        hipLaunchKernelGGL(where_am_i, dim3(32768), dim3(1024), 0, w, d_out2, 200000);   // ~ 128 waves of 256 WGs x 0.1 ms
        CHECK(hipEventRecord(b, w));
        CHECK(hipEventRecord(c, ch));
        hipLaunchKernelGGL(where_am_i, dim3(1), dim3(1024), 0, ch, d_out, 200000);
        CHECK(hipEventRecord(d, ch));
        CHECK(hipDeviceSynchronize());
        float tw, tc;
        CHECK(hipEventElapsedTime(&tw, a, b));
        CHECK(hipEventElapsedTime(&tc, c, d));
        printf("%-52s: wide kernel %.2f ms; 1-work-group kernel enqueued right behind it on another stream took %.3f ms (alone: ~0.1)\n",
               names[variant], tw, tc);
    }
    // software reservation: persistent 768-work-group grid, work-groups on CU0/SE0/SH0 retire at once
    for (int reserve = 0; reserve < 2; ++reserve) {
        hipStream_t w, ch;
        CHECK(hipStreamCreate(&w)); CHECK(hipStreamCreate(&ch));
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(a, w));
        hipLaunchKernelGGL(persistent_spin, dim3(768), dim3(256), 0, w, d_out2, 12000000LL, reserve);   // ~5 ms
        CHECK(hipEventRecord(b, w));
        CHECK(hipEventRecord(c, ch));
        CHECK(hipFuncSetAttribute((const void*)whole_cu, hipFuncAttributeMaxDynamicSharedMemorySize, 160000));
        hipLaunchKernelGGL(whole_cu, dim3(1), dim3(1024), 160000, ch, d_out, 200000);
        CHECK(hipEventRecord(d, ch));
        CHECK(hipDeviceSynchronize());
        float tw, tc;
        CHECK(hipEventElapsedTime(&tw, a, b));
        CHECK(hipEventElapsedTime(&tc, c, d));
        uint32_t h[2];
        CHECK(hipMemcpy(h, d_out, 8, hipMemcpyDeviceToHost));
        printf("persistent 768-work-group grid, reserve=%d: %.2f ms; 1024-thread / 160-KB-LDS work-group on another stream: %.3f ms, ran on x%u.se%u.sh%u.cu%u\n",
               reserve, tw, tc, h[0] & 0xf, (h[1] >> 13) & 7, (h[1] >> 12) & 1, (h[1] >> 8) & 0xf);
    }
    return 0;
}
