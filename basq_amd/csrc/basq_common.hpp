// basq_common.hpp -- what the three translation units of libbasq_hip.so share (gfx950 / MI355X / CDNA4):
// includes, launch check, wave-level reductions, lane reads, the correctly rounded quotient.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/basq_hip.h"
#

typedef double d4 __attribute__((ext_vector_type(4)));

#define BASQ_CHECK_LAUNCH()                                   \
    do {                                                      \
        if (hipGetLastError() != hipSuccess) return BASQ_ELAUNCH; \
    } while (0)


#ifndef BASQ_CAR_THREADS
#define BASQ_CAR_THREADS 1024
#endif
#ifndef BASQ_CHOL_THREADS
#define BASQ_CHOL_THREADS 1024   // work-group size of chol_inv_lds_kernel (multiple of 128)
#endif

__device__ __forceinline__ double sum16(double v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    return v;
}

#ifndef BASQ_WAVE_SUM_DPP
#define BASQ_WAVE_SUM_DPP 1
#endif
// v shifted across lanes by a DPP control (row_shr:n = 0x110+n, row_bcast:15 = 0x142, row_bcast:31 = 0x143);
// lanes without a source (or masked off) receive 0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_shift_f64(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, ROW_MASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// Sum over the 64 lanes, same value returned to every lane, fixed association.  DPP form: prefix sums inside
// each row of 16 lanes (row_shr 1,2,4,8), row totals forwarded (row_bcast 15 / 31), lane 63 read back through
// an SGPR -- ~20 VALU instructions instead of six dependent ds_bpermute round trips.
__device__ __forceinline__ double wave_sum(double v) {
#if BASQ_WAVE_SUM_DPP
    v += dpp_shift_f64<0x111, 0xf>(v);
    v += dpp_shift_f64<0x112, 0xf>(v);
    v += dpp_shift_f64<0x114, 0xf>(v);
    v += dpp_shift_f64<0x118, 0xf>(v);
    v += dpp_shift_f64<0x142, 0xa>(v);
    v += dpp_shift_f64<0x143, 0xc>(v);
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)b, 63);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
#else
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
#endif
}

// optional in-kernel phase stamps (tools/ns_prof.hip): wave w, lane 0 -> g_ns_prof[(t * 8 + slot) * 16 + w]
#ifdef BASQ_NS_PROF
__device__ long long* g_ns_prof;
#define BASQ_NS_STAMP(t, slot)                                                                     \
    do {                                                                                           \
        if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) g_ns_prof[((t) * 8 + (slot)) * 16 + (threadIdx.x >> 6)] = clock64(); \
    } while (0)
#else
#define BASQ_NS_STAMP(t, slot) do { } while (0)
#endif


// Correctly rounded quotient a / b from rb = RN(1/b) with two FMAs (Markstein 1990: q0 = RN(a rb),
// r = a - b q0 exactly (FMA), q = RN(q0 + r rb) = RN(a/b) for normal, finite operands).  The elimination
// performs ~1e6 divisions by the SAME pivot per step; this keeps the reference's rounding at 1/6 of the cost.
__device__ __forceinline__ double div_by_recip(double a, double b, double rb) {
    const double q0 = a * rb;
    const double r = __builtin_fma(-b, q0, a);
    return __builtin_fma(r, rb, q0);
}

// value of lane `src` (wave-uniform index) delivered through SGPRs
__device__ __forceinline__ double readlane_f64(double v, int src) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)b, src);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// v moved across lanes by a DPP control (row_shr:n = 0x110+n, row_bcast:15 = 0x142, row_bcast:31 = 0x143);
// lanes without a source, or in rows masked off, receive `fill`.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_shift_fill_f64(double v, double fill) {
    const long long b = __double_as_longlong(v), o = __double_as_longlong(fill);
    const int lo = __builtin_amdgcn_update_dpp((int)o, (int)b, CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(o >> 32), (int)(b >> 32), CTRL, ROW_MASK, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

typedef __attribute__((address_space(1))) unsigned long long basq_gu64;
typedef __attribute__((address_space(1))) unsigned int basq_gu32;
#define BASQ_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
#define BASQ_SPIN_LIMIT (1u << 22)       // polls (with s_sleep) before a cluster kernel gives up: ~0.3 s
#define BASQ_ABORT_COUNT 0x40000000

// Newton steps behind v_rcp_f64 / v_rsq_f64.  The seeds are good to 2^-24 (measured: 2.5e8 ulp), so TWO steps reach the rounding
// floor -- sqrt 1.4 ulp, 1/sqrt 1.6 ulp, 1/d 0.5 ulp over 4M values; a third changes nothing (1.9 / 1.8 / 0.5:
// tools/newton_probe.hip, profiles/r07_b_newton_steps_after_rsq_rcp.txt) and sits on the serial chain of every reflector.
#define BASQ_NEWTON_STEPS 2
// 1/d to ~1 ulp without the scaling / fix-up of an IEEE divide (d is a normal, finite reflector norm here)
__device__ __forceinline__ double recip_nr(double d) {
    double y = __builtin_amdgcn_rcp(d);
#pragma unroll
    for (int it = 0; it < BASQ_NEWTON_STEPS; ++it) y = __builtin_fma(__builtin_fma(-d, y, 1.0), y, y);
    return y;
}

// sqrt(d) and 1/sqrt(d) to ~1.5 ulp: v_rsq_f64 seed + coupled Newton steps (g -> sqrt(d), h -> 1/(2 sqrt(d))).
__device__ __forceinline__ void sqrt_rsqrt_nr(double d, double& root, double& rroot) {
    const double y = __builtin_amdgcn_rsq(d);
    double g = d * y, h = 0.5 * y;
#pragma unroll
    for (int it = 0; it < BASQ_NEWTON_STEPS; ++it) {
        const double r = __builtin_fma(-h, g, 0.5);
        g = __builtin_fma(g, r, g);
        h = __builtin_fma(h, r, h);
    }
    root = g;
    rroot = 2.0 * h;
}
__device__ __forceinline__ double rsqrt_nr(double d) {
    double g, r;
    sqrt_rsqrt_nr(d, g, r);
    return r;
}

