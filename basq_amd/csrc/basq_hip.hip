// basq_hip.hip -- gfx950 (MI355X, CDNA4) kernels behind include/basq_hip.h.
//
// Hot path: kernel recombination of ma921/BASQ (BASQ/_rchq.py).  Everything is float64 (SURVEY §8c:
// the reference's index selection is only well-posed in fp64).  Wave = 64 lanes; the pairwise
// exponent arguments are produced on the f64 matrix cores (v_mfma_f64_16x16x4_f64) from packed
// operands, the transcendental epilogue runs on the fp64 VALU, which is the binding unit.
//
// MFMA f64 16x16x4 lane maps (cdna_hip_programming.md §3):  lane l, c = l & 15, g = l >> 4
//     A operand: A[row = c][k = g]      B operand: B[k = g][col = c]
//     C/D:       D[reg r] = D[row = g + 4 r][col = c]
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/basq_hip.h"
#include "exp_coeffs.inc"

typedef double d4 __attribute__((ext_vector_type(4)));

#define BASQ_CHECK_LAUNCH()                                   \
    do {                                                      \
        if (hipGetLastError() != hipSuccess) return BASQ_ELAUNCH; \
    } while (0)

// ------------------------------------------------------------------------------------------------
// fp64 exp for arguments <= 0 (every kernel family evaluates exp of a non-positive number).
// x = n ln2 + r, |r| <= ln2/2; degree-10 polynomial (2.9e-16 rel. before rounding); 2^n applied by an
// integer add on the exponent field; arguments below about -708 return (almost exactly) 0.  Valid for
// -1.4e9 < x <= ~1 (the callers pass minus a scaled squared distance or minus a scaled distance).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double exp_nonpos(double x) {
    const double MAGIC = 0x1.8p52;
    const double t = __builtin_fma(x, BASQ_LOG2E, MAGIC);
    const double nf = t - MAGIC;
    double r = __builtin_fma(nf, -BASQ_LN2_HI, x);
    r = __builtin_fma(nf, -BASQ_LN2_LO, r);
    double p = BASQ_EXP_P10;
    p = __builtin_fma(p, r, BASQ_EXP_P9);
    p = __builtin_fma(p, r, BASQ_EXP_P8);
    p = __builtin_fma(p, r, BASQ_EXP_P7);
    p = __builtin_fma(p, r, BASQ_EXP_P6);
    p = __builtin_fma(p, r, BASQ_EXP_P5);
    p = __builtin_fma(p, r, BASQ_EXP_P4);
    p = __builtin_fma(p, r, BASQ_EXP_P3);
    p = __builtin_fma(p, r, BASQ_EXP_P2);
    p = __builtin_fma(p, r, BASQ_EXP_P1);
    p = __builtin_fma(p, r, BASQ_EXP_P0);
    const int n = __double2loint(t);               // low word of t holds n (two's complement)
    int hi = __double2hiint(p) + (n << 20);        // p * 2^n, p in [0.70, 1.42]
    hi = (n < -1021) ? 0 : hi;                     // underflow: return ~0 (a denormal <= 2^-1042)
    return __hiloint2double(hi, __double2loint(p));
}

// Block-sum form of exp: x = (2048 n + j) ln2/2048 + r, T[j] = 2^(j/2048) from a 16-KB LDS table, degree-3 polynomial
// in r (|r| < 1.7e-4, error 1e-17), 2^n by ldexp -- 8 fp64 VALU instructions + one ds_read, no fp64 transcendental
// hardware involved.  (Round 1 A/B-timed three other evaluations -- a degree-10 polynomial without a table, a 32-entry
// table in LDS or in global memory with a degree-5 polynomial; profiles/r01_blocksum_exp_modes_ab.txt -- this one won.)
// The constants are pinned in VGPRs: an fp64 literal costs an SGPR pair and a constant-bus slot per use.
// Scheme 2 (template parameter XS of the block-sum kernel): 4096-entry table (32 KB of LDS per work-group), degree-2
// interpolant in r (|r| < 8.5e-5, max relative error 2.5e-14) -- one fp64 instruction fewer per kernel value: 7.47 vs 7.79 ms
// per 1e10 pairs (profiles/r03_n_exp_scheme_ab.txt).  Used by the recombination's block sums, whose selection is stable under
// kernel perturbations up to 1e-7 (SURVEY finding 3); the kernel mat-vec (GP means, Gaussian moments: sums with
// cancellation) and the squared-covariance sums keep scheme 1 (1e-17).
__device__ const double basq_exp_tab2048_g[2048] = BASQ_EXP_TAB2048;
__device__ const double basq_exp_tab4096_g[4096] = BASQ_EXP_TAB4096;

#ifndef BASQ_BLOCKSUM_EXP_SCHEME
#define BASQ_BLOCKSUM_EXP_SCHEME 2      // the recombination's block sums (A/B builds: -DBASQ_BLOCKSUM_EXP_SCHEME=1)
#endif

template <int XS>
struct ExpScheme {
    static constexpr int N = (XS == 2) ? 4096 : 2048;
    static constexpr int SHIFT = (XS == 2) ? 12 : 11;
};

struct ExpK {
    double k32, nhi, magic, c3, c2, one;
};

__device__ __forceinline__ double vgpr_const(double x) {
    asm volatile("" : "+v"(x));
    return x;
}

template <int XS>
__device__ __forceinline__ void expk_init(ExpK& k) {
    k.magic = vgpr_const(0x1.8p52);
    if (XS == 2) {
        k.k32 = vgpr_const(BASQ_4096_OVER_LN2);
        k.nhi = vgpr_const(-BASQ_LN2_4096_HI);
        k.c3 = vgpr_const(BASQ_EXP_V2);
        k.c2 = vgpr_const(BASQ_EXP_V1);
        k.one = vgpr_const(BASQ_EXP_V0);
    } else {
        k.k32 = vgpr_const(BASQ_2048_OVER_LN2);
        k.nhi = vgpr_const(-BASQ_LN2_2048_HI);
        k.c3 = vgpr_const(BASQ_EXP_U3);
        k.c2 = vgpr_const(BASQ_EXP_U2);
        k.one = vgpr_const(1.0);
    }
}

// `tab` = LDS copy of the table (exp_table_init); valid for -1.4e9 < x <= ~1, exact 0 below ~-745.
template <int XS>
__device__ __forceinline__ double exp_nonpos_k(double x, const ExpK& k, const double* tab) {
    const double t = __builtin_fma(x, k.k32, k.magic);
    const int ti = __double2loint(t);                 // N n + j  (two's complement)
    const double T = tab[ti & (ExpScheme<XS>::N - 1)];
    const double nf = t - k.magic;
    // ln2/N is used as ONE correctly rounded constant; the dropped tail |nf| * 1.1e-20 is a relative error of
    // < 2e-15 in the kernel value for every argument whose exp exceeds 1e-22 (|nf| < 1.5e5)
    const double r = __builtin_fma(nf, k.nhi, x);
    double w = __builtin_fma(k.c3, r, k.c2);
    w = __builtin_fma(w, r, k.one);
    const double e = (XS == 2) ? (T * w)                          // T (v0 + v1 r + v2 r^2)
                               : __builtin_fma(T * r, w, T);      // T (1 + r w)
    return ldexp(e, ti >> ExpScheme<XS>::SHIFT);
}

template <int XS>
__device__ __forceinline__ void exp_table_init(double* tab) {
    const double* src = (XS == 2) ? basq_exp_tab4096_g : basq_exp_tab2048_g;
    for (int i = threadIdx.x; i < ExpScheme<XS>::N; i += blockDim.x) tab[i] = src[i];
    __syncthreads();
}

// Kernel value (without outputscale) from D = -1/2 |(x-y)/l|^2.
template <int FAM>
__device__ __forceinline__ double kernel_from_arg(double D) {
    if (FAM == BASQ_FAMILY_RBF) {
        return exp_nonpos(D);                      // D <= ~1e-14: no clamp needed for exp
    } else {
        const double r2 = fmax(-2.0 * D, 1e-30);   // gpytorch: clamp_min(1e-30) before sqrt
        const double r = sqrt(r2);
        if (FAM == BASQ_FAMILY_MATERN52) {
            const double a = 0x1.1e3779b97f4a8p+1 * r;   // sqrt(5) r
            const double poly = (a + 1.0) + (5.0 / 3.0) * r2;
            return poly * exp_nonpos(-a);
        } else {
            const double a = 0x1.bb67ae8584caap+0 * r;   // sqrt(3) r
            return (a + 1.0) * exp_nonpos(-a);
        }
    }
}

// sqrt(x) for x in [1e-30, 1e300) -- the clamped squared distance of the Matern kernels -- without the library routine's
// scaling and special-case handling (two v_ldexp_f64, a v_cmp_class_f64 and four selects per value: 16 instructions, where
// the block sums are bound by their instruction count): v_rsq_f64 + the coupled Newton step for (sqrt, 1/(2 sqrt)) + one
// residual correction, 8 instructions, error <= 1 ulp (round 4; the Gram / mat-vec kernels keep the library sqrt).
__device__ __forceinline__ double sqrt_pos_fast(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    const double d = __builtin_fma(-g, g, x);
    return __builtin_fma(d, h, g);
}

#ifndef BASQ_FAST_SQRT
#define BASQ_FAST_SQRT 1        // A/B builds: -DBASQ_FAST_SQRT=0 (library sqrt in the block sums)
#endif
template <int FAM, int XS>
__device__ __forceinline__ double kernel_from_arg_k(double D, const ExpK& k, const double* tab) {
    if (FAM == BASQ_FAMILY_RBF) {
        return exp_nonpos_k<XS>(D, k, tab);
    } else {
        const double r2 = fmax(-2.0 * D, 1e-30);
        const double r = BASQ_FAST_SQRT ? sqrt_pos_fast(r2) : sqrt(r2);
        if (FAM == BASQ_FAMILY_MATERN52) {
            const double a = 0x1.1e3779b97f4a8p+1 * r;
            const double poly = (a + 1.0) + (5.0 / 3.0) * r2;
            return poly * exp_nonpos_k<XS>(-a, k, tab);
        } else {
            const double a = 0x1.bb67ae8584caap+0 * r;
            return (a + 1.0) * exp_nonpos_k<XS>(-a, k, tab);
        }
    }
}

// Sum over the 16 lanes that share g = lane >> 4 (the 16 columns of an MFMA tile); fixed butterfly
// order, result in every lane of the group.
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

// ------------------------------------------------------------------------------------------------
// pack / init / mean
// ------------------------------------------------------------------------------------------------
__global__ void col_mean_kernel(const double* __restrict__ X, long long n, int d, double* __restrict__ mean) {
    // one block; thread (k, lane-in-column) ; fixed-order two-level sum => deterministic
    __shared__ double part[1024];
    const int per = blockDim.x / d;                 // threads per column
    const int k = threadIdx.x % d, t = threadIdx.x / d;
    double acc = 0.0;
    if (t < per) {
        // eight loads in flight, added in the same order (a plain loop waits for every load before it issues the next one)
        long long i = t;
        for (; i + 7 * (long long)per < n; i += 8 * (long long)per) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = X[(i + u * (long long)per) * d + k];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; i < n; i += per) acc += X[i * d + k];
    }
    part[threadIdx.x] = (t < per) ? acc : 0.0;
    __syncthreads();
    if (threadIdx.x < d) {
        double s = 0.0;
        for (int u = 0; u < per; ++u) s += part[u * d + threadIdx.x];
        mean[threadIdx.x] = n > 0 ? s / (double)n : 0.0;
    }
}

// One thread per point for the arithmetic (the norm h is accumulated in coordinate order, as before), but the rows move
// through an LDS tile: a point's d inputs / kp outputs are 80 / 96 bytes at d = 10, so per-thread row accesses touch every
// cache line 8-12 times from different lanes (189 us for the 1e6-candidate pack = a quarter of the HBM rate); the tile
// is read and written with consecutive lanes on consecutive doubles instead.
__global__ void __launch_bounds__(256) pack_points_kernel(const double* __restrict__ X, long long n, int d, int kp,
                                                          const double* __restrict__ center, double inv_ell, int role,
                                                          double* __restrict__ out, int ppb) {
    extern __shared__ double tile[];                       // [ppb][kp | 1]: odd stride -> a thread's row walk hits distinct banks
    const int ld = kp | 1;
    const long long i0 = (long long)blockIdx.x * ppb;      // ppb <= 256 points per block (128 for rows of more than 28 doubles)
    const int cnt = (n - i0 < ppb) ? (int)(n - i0) : ppb;
    const double* src = X + i0 * d;
    for (int e = threadIdx.x; e < cnt * d; e += 256) tile[(e / d) * ld + (e % d)] = src[e];
    __syncthreads();
    if (threadIdx.x < cnt) {
        double* o = tile + threadIdx.x * ld;
        double h = 0.0;
        for (int k = 0; k < d; ++k) {
            const double c = center ? center[k] : 0.0;
            const double v = (o[k] - c) * inv_ell;
            o[k] = v;
            h = __builtin_fma(v, v, h);
        }
        h *= -0.5;
        for (int k = d; k < kp - 2; ++k) o[k] = 0.0;
        o[kp - 2] = (role == BASQ_ROLE_A) ? h : 1.0;
        o[kp - 1] = (role == BASQ_ROLE_A) ? 1.0 : h;
    }
    __syncthreads();
    double* dst = out + i0 * kp;
    for (int e = threadIdx.x; e < cnt * kp; e += 256) dst[e] = tile[(e / kp) * ld + (e % kp)];
}

__global__ void init_state_kernel(double* __restrict__ mu, long long* __restrict__ gid, long long Rl, long long gid0,
                                  double w0) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Rl) return;
    mu[i] = w0;
    gid[i] = gid0 + i;
}

// ------------------------------------------------------------------------------------------------
// Fused block sums.  One wave owns a 64 x 16 tile of (Nystrom rows x sets):
// the A fragments of its 64 rows stay in registers for the whole launch; every iteration streams
// the 16 candidates of one block that fall in the wave's 16 sets (B fragments), issues
// JT*KK MFMAs for the exponent arguments and evaluates 16 kernel values per lane on the VALU.
//
// Why the matrix cores for d-dimensional distances (measured, profiles/r02_microbench_fp64_rates.txt, in-kernel clock
// 2.3-2.4 GHz): v_mfma_f64_16x16x4 issues once per ~106 cycles (47 TF/s), a v_fma_f64 with constant operands once per
// 4.2 (73 TF/s) -- but a distance FMA reads THREE vector registers, and all-VALU forms of this kernel (candidates through
// the scalar cache in round 1, broadcast from an LDS tile in round 2: 18 tunings, profiles/r02_blocksum_lds_form_sweep.txt)
// never got below 11.5 ms per 1e10 pairs against 8.4-9.1 ms for this form, whose operands arrive as plain coalesced
// vector loads.  Per 64 pairs at d = 10: 0.78 MFMA (83 cycles) + 14.4 VALU (60 cycles) = its measured 143 cycles.
// ------------------------------------------------------------------------------------------------
struct BlocksumArgs {
    const double* nys;
    const double* cand;
    const double* mu;
    const double* wx;
    double* Xpart;
    double* totpart;
    long long Rl, off, n_full;
    long long blk_lo, blk_hi, blk_per_chunk;   // global block-index range touched by this rank
    int class_mod, class0;                     // > 0: chunk c = the blocks b with b % class_mod == class0 + c (residue classes)
    int m, S, n_chunks, n_stiles, n_jgroups;
    // Device-resident round descriptor (basq_round_next_i64): when set, the candidate range comes from HBM instead of the
    // launch arguments, so the host can enqueue a round before it knows how many candidates survived the previous one.
    const long long* geo;                      // {R, n_full, reg_hi, violation, nb, n_tail, -, -}
    int geo_mode;                              // 1: positions [0, reg_hi)   2: [reg_hi, R)   3: [0, R)   4: the remainder [n_full, R) as a block of its own
};

// Candidate range of a descriptor-driven launch (wave-uniform scalar loads and arithmetic; the formulas of blocksum_impl).
// The descriptor carries the round's GLOBAL geometry and this rank's shard [off, off + Rl) of the live positions
// (geo[6], geo[7]; one rank: [0, R)); the launch covers the intersection of the shard with the mode's position range.
template <int KP>
__device__ __forceinline__ void blocksum_apply_geo(BlocksumArgs& A) {
    const long long R = A.geo[0], n_full = A.geo[1], reg_hi = A.geo[2];
    const long long s_off = A.geo[6], s_end = A.geo[6] + A.geo[7];
    long long lo = 0, hi = R;
    if (A.geo_mode == 1) hi = reg_hi;
    else if (A.geo_mode == 2) lo = reg_hi;
    else if (A.geo_mode == 4) lo = n_full;
    if (lo < s_off) lo = s_off;
    if (hi > s_end) hi = s_end;
    if (hi < lo) hi = lo;
    const long long skip = lo - s_off;                        // local index of the first position of the launch
    A.cand += skip * KP;
    A.mu += skip;
    if (A.wx) A.wx += skip;
    long long off = lo;
    const long long Rl = hi - lo;
    long long nf = n_full;
    if (A.geo_mode == 4) {                                    // the ragged remainder as ONE block of its own: point k in set k
        off = lo - n_full;                                    // (SOBER/_rchq.py:127-135; fewer than S points)
        nf = A.S;
    }
    A.off = off;
    A.Rl = Rl;
    A.n_full = nf;
    const long long n_full_eff = nf;
    const long long lim = (off + Rl < n_full_eff) ? (off + Rl) : n_full_eff;
    if (lim > off) {
        A.blk_lo = off / A.S;
        A.blk_hi = (lim + A.S - 1) / A.S;
    } else {
        A.blk_lo = 0;
        A.blk_hi = 0;
    }
    A.blk_per_chunk = (A.blk_hi - A.blk_lo + A.n_chunks - 1) / A.n_chunks;
    if (A.blk_per_chunk < 1) A.blk_per_chunk = 1;
}

template <int KK>
struct CandFrag {
    double b[KK];
    double w;    // kernel weight mu * wx (0 for masked columns)
    double wm;   // mu (0 for masked columns)
};

template <int KK>
__device__ __forceinline__ void load_cand(CandFrag<KK>& f, const BlocksumArgs& A, long long pl, bool ok, int g) {
    constexpr int KP = KK * 4;
    const long long row = ok ? pl : 0;
    const double* src = A.cand + row * KP + g;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) f.b[kk] = src[kk * 4];
    const double m_ = A.mu[row];
    const double x_ = A.wx ? A.wx[row] : 1.0;
    f.wm = ok ? m_ : 0.0;
    f.w = ok ? m_ * x_ : 0.0;
}

template <int KK, int FAM, int JT, int XS>
__device__ __forceinline__ void tile_accumulate(const double (&a)[JT][KK], const CandFrag<KK>& f, double (&acc)[JT][4],
                                                const ExpK& ek, const double* tab) {
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) {
        d4 D = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) D = __builtin_amdgcn_mfma_f64_16x16x4f64(a[jt][kk], f.b[kk], D, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[jt][r] = __builtin_fma(kernel_from_arg_k<FAM, XS>(D[r], ek, tab), f.w, acc[jt][r]);
    }
}

// Blocks of prefetch distance for the candidate rows of the block sums: 2 while the third fragment keeps the kernel at
// three waves per SIMD (<= 170 registers: KP <= 32), 1 for the wider rows (KK = 9, 10: 214-222 registers with a third
// fragment).  Measured at the headline shape: 6.36 -> 6.20 ms per 14-class launch (profiles/r04_k_blocksum_prefetch2.txt).
#ifndef BASQ_BS_PREFETCH
#define BASQ_BS_PREFETCH 0      // 0 = by KK (A/B builds: -DBASQ_BS_PREFETCH=1 or 2)
#endif
#define BASQ_BS_PF_FOR(KK) (BASQ_BS_PREFETCH ? BASQ_BS_PREFETCH : ((KK) <= 8 ? 2 : 1))
#ifndef BASQ_BS_WAVES
#define BASQ_BS_ATTR
#else
#define BASQ_BS_ATTR __attribute__((amdgpu_waves_per_eu(BASQ_BS_WAVES, BASQ_BS_WAVES)))
#endif
template <int KK, int FAM, int JT, int XS>
__global__ void __launch_bounds__(256) BASQ_BS_ATTR blocksum_kernel(const BlocksumArgs A_in) {
    constexpr int KP = KK * 4;
    BlocksumArgs A = A_in;
    if (A.geo) blocksum_apply_geo<KP>(A);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    // XCD-aware work-group -> tile map.  Work-groups are dealt round-robin over the 8 XCDs (blockIdx.x % 8), each with
    // its own 4-MB L2.  All row groups of one (set tile, chunk) pair read the SAME candidate slice (~0.5 MB at the
    // headline size), so they are given to ONE XCD, consecutively: the slice is fetched into that L2 once instead of
    // once per row group (PMC: 1.7 GB -> 94 MB per round-1 launch, profiles/r02_traffic.json).  Placement only affects
    // speed, never results.
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int jg = seq % A.n_jgroups;
    const int gidx = (seq / A.n_jgroups) * 8 + xcd;                 // (set tile, chunk) pair of this work-group
    const int st = gidx % A.n_stiles;
    const int chunk = gidx / A.n_stiles;
    const int j0 = (jg * 4 + wave) * (16 * JT);
    __shared__ double exp_tab[ExpScheme<XS>::N];
    exp_table_init<XS>(exp_tab);   // the only barrier of this kernel, before any early exit
    if (chunk >= A.n_chunks) return;   // padding work-groups of the last group of 8 pairs (work-group uniform)
    if (j0 >= A.m) return;     // wave-uniform
    const int s0 = st * 16;
    ExpK ek;
    expk_init<XS>(ek);

    double a[JT][KK];
#pragma unroll
    for (int jt = 0; jt < JT; ++jt)
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) a[jt][kk] = A.nys[(long long)(j0 + jt * 16 + c) * KP + kk * 4 + g];

    double acc[JT][4];
#pragma unroll
    for (int jt = 0; jt < JT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[jt][r] = 0.0;
    double tot = 0.0;

    const bool col_ok = (s0 + c) < A.S;
    const long long lim = (A.off + A.Rl < A.n_full) ? (A.off + A.Rl) : A.n_full;   // end of block positions held here
    // blocks of this chunk: a contiguous range (step 1), or -- residue-class mode -- every class_mod-th block of the
    // rank's range, starting at the first block congruent to this chunk's class
    long long bA, bB, bstep;
    if (A.class_mod > 0) {
        bstep = A.class_mod;
        const long long cls = A.class0 + chunk;
        bA = A.blk_lo + ((cls - A.blk_lo % bstep) % bstep + bstep) % bstep;
        bB = A.blk_hi;
    } else {
        bstep = 1;
        bA = A.blk_lo + (long long)chunk * A.blk_per_chunk;
        bB = bA + A.blk_per_chunk;
        if (bB > A.blk_hi) bB = A.blk_hi;
    }

    if (bA < bB) {
        // Blocks whose 16 columns all lie inside this rank's block positions take the fast path: the lane's
        // row pointer just advances by bstep * S rows per block (no masks, no 64-bit index arithmetic).  The (at most
        // two) edge blocks before / after that wave-uniform range use the masked path.
        const long long first_ok = (A.off - s0 + A.S - 1) / A.S;                 // smallest i with i*S + s0 >= off
        const long long last_ok = (lim - s0 - 16 >= 0) ? ((lim - s0 - 16) / A.S) : -1;   // largest i with i*S+s0+15 < lim
        const bool tile_full = (s0 + 16) <= A.S;
        long long i = bA;
        // masked prologue blocks
        for (; i < bB && (!tile_full || i < first_ok); i += bstep) {
            const long long pg = i * A.S + s0 + c;
            CandFrag<KK> f;
            load_cand<KK>(f, A, pg - A.off, col_ok && pg >= A.off && pg < lim, g);
            tile_accumulate<KK, FAM, JT, XS>(a, f, acc, ek, exp_tab);
            tot += f.wm;
        }
        const long long bF1 = (last_ok + 1 < bB) ? (last_ok + 1) : bB;           // end of the fast range
        if (tile_full && i < bF1) {
            const long long p0 = i * A.S + s0 + c - A.off;                       // local row of this lane, block i
            const double* rp = A.cand + p0 * KP + g;
            const double* mp = A.mu + p0;
            const double* xp = A.wx ? (A.wx + p0) : nullptr;
            const long long rstep = bstep * (long long)A.S * KP, mstep = bstep * (long long)A.S;
            if constexpr (BASQ_BS_PF_FOR(KK) == 2) {
                // candidate rows requested TWO blocks ahead: every row is read once, from HBM or the far L2, and a block of work
                // (~1 us with three waves per SIMD) does not always cover that; a third fragment costs 10 registers
                CandFrag<KK> cur, nxt, nx2;
                auto fetch = [&](CandFrag<KK>& f, const double* r, const double* mq, const double* xq) {
#pragma unroll
                    for (int kk = 0; kk < KK; ++kk) f.b[kk] = r[kk * 4];
                    f.wm = mq[0];
                    f.w = xq ? f.wm * xq[0] : f.wm;
                };
                const long long nblk = (bF1 - i + bstep - 1) / bstep;               // blocks of the fast range
                fetch(cur, rp, mp, xp);
                const long long o1 = (nblk > 1) ? 1 : 0;
                fetch(nxt, rp + o1 * rstep, mp + o1 * mstep, xp ? xp + o1 * mstep : nullptr);
                for (long long t = 0; t < nblk; ++t) {
                    const long long o2 = (t + 2 < nblk) ? (t + 2) : (nblk - 1);     // the last trips re-read the last row
                    fetch(nx2, rp + o2 * rstep, mp + o2 * mstep, xp ? xp + o2 * mstep : nullptr);
                    tile_accumulate<KK, FAM, JT, XS>(a, cur, acc, ek, exp_tab);
                    tot += cur.wm;
                    cur = nxt;
                    nxt = nx2;
                }
                i += nblk * bstep;
            } else {
                CandFrag<KK> cur, nxt;
#pragma unroll
                for (int kk = 0; kk < KK; ++kk) cur.b[kk] = rp[kk * 4];
                cur.wm = mp[0];
                cur.w = xp ? cur.wm * xp[0] : cur.wm;
                for (; i < bF1; i += bstep) {
                    const bool more = (i + bstep < bF1);
                    const double* rn = more ? (rp + rstep) : rp;                     // last trip re-reads its own row
                    const double* mn = more ? (mp + mstep) : mp;
#pragma unroll
                    for (int kk = 0; kk < KK; ++kk) nxt.b[kk] = rn[kk * 4];
                    nxt.wm = mn[0];
                    if (xp) {
                        const double* xn = more ? (xp + mstep) : xp;
                        nxt.w = nxt.wm * xn[0];
                        xp = xn;
                    } else {
                        nxt.w = nxt.wm;
                    }
                    tile_accumulate<KK, FAM, JT, XS>(a, cur, acc, ek, exp_tab);
                    tot += cur.wm;
                    cur = nxt;
                    rp = rn;
                    mp = mn;
                }
            }
        }
        // masked epilogue blocks
        for (; i < bB; i += bstep) {
            const long long pg = i * A.S + s0 + c;
            CandFrag<KK> f;
            load_cand<KK>(f, A, pg - A.off, col_ok && pg >= A.off && pg < lim, g);
            tile_accumulate<KK, FAM, JT, XS>(a, f, acc, ek, exp_tab);
            tot += f.wm;
        }
    }

    // Ragged tail (positions >= n_full all belong to set S-1, BASQ/_rchq.py:91-99): 16 tail candidates
    // per iteration, one per column; folded into the column that owns set S-1 at the end.
    const long long t0 = (A.n_full > A.off) ? (A.n_full - A.off) : 0;   // first local tail position
    if (chunk == A.n_chunks - 1 && st == A.n_stiles - 1 && t0 < A.Rl) {
        double tacc[JT][4];
#pragma unroll
        for (int jt = 0; jt < JT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) tacc[jt][r] = 0.0;
        double ttot = 0.0;
        for (long long p = t0; p < A.Rl; p += 16) {
            CandFrag<KK> f;
            load_cand<KK>(f, A, p + c, (p + c) < A.Rl, g);
            tile_accumulate<KK, FAM, JT, XS>(a, f, tacc, ek, exp_tab);
            ttot += f.wm;
        }
        const int c_last = (A.S - 1) - s0;
#pragma unroll
        for (int jt = 0; jt < JT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double v = sum16(tacc[jt][r]);
                if (c == c_last) acc[jt][r] += v;
            }
        const double tv = sum16(ttot);
        if (c == c_last) tot += tv;
    }

    if (col_ok) {
        double* out = A.Xpart + (long long)chunk * A.m * A.S;
#pragma unroll
        for (int jt = 0; jt < JT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = j0 + jt * 16 + g + 4 * r;
                if (j < A.m) out[(long long)j * A.S + s0 + c] = acc[jt][r];
            }
        if (A.totpart && jg == 0 && wave == 0 && g == 0) A.totpart[(long long)chunk * A.S + s0 + c] = tot;
    }
}

// Row tiles per wave: 4 (64 Nystrom rows) while the A fragments fit comfortably; 2 for KP >= 24 (d >= 21), where
// 4 x KP/4 fragment registers would push the kernel to one wave per SIMD.  basq_amd/_partition.py mirrors this.
#ifndef BASQ_JT_SMALL
#define BASQ_JT_SMALL 4         // row tiles per wave for KP <= 20 (A/B builds: -DBASQ_JT_SMALL=2)
#endif
#define BASQ_JT_FOR(KK) ((KK) >= 6 ? 2 : BASQ_JT_SMALL)

template <int KK, int FAM, int XS>
static int launch_blocksum(const BlocksumArgs& A, hipStream_t st) {
    constexpr int JT = BASQ_JT_FOR(KK);
    BlocksumArgs B = A;
    B.n_jgroups = (A.m + 64 * JT - 1) / (64 * JT);         // 4 waves x 16*JT rows per block
    const long long npairs = (long long)A.n_stiles * A.n_chunks;
    const long long nblk = ((npairs + 7) / 8) * 8 * B.n_jgroups;   // (set tile, chunk) pairs padded to the 8 XCDs
    if (nblk <= 0 || nblk > 0x7fffffffLL) return BASQ_EINVAL;
    hipLaunchKernelGGL((blocksum_kernel<KK, FAM, JT, XS>), dim3((unsigned)nblk), dim3(256), 0, st, B);
    BASQ_CHECK_LAUNCH();
    return BASQ_OK;
}

template <int KK>
static int dispatch_blocksum_fam(int fam, const BlocksumArgs& A, hipStream_t st, int xs) {
    if (xs == 2) {
        switch (fam) {
            case BASQ_FAMILY_RBF: return launch_blocksum<KK, BASQ_FAMILY_RBF, 2>(A, st);
            case BASQ_FAMILY_MATERN52: return launch_blocksum<KK, BASQ_FAMILY_MATERN52, 2>(A, st);
            case BASQ_FAMILY_MATERN32: return launch_blocksum<KK, BASQ_FAMILY_MATERN32, 2>(A, st);
        }
        return BASQ_EUNSUPPORTED;
    }
    switch (fam) {
        case BASQ_FAMILY_RBF: return launch_blocksum<KK, BASQ_FAMILY_RBF, 1>(A, st);
        case BASQ_FAMILY_MATERN52: return launch_blocksum<KK, BASQ_FAMILY_MATERN52, 1>(A, st);
        case BASQ_FAMILY_MATERN32: return launch_blocksum<KK, BASQ_FAMILY_MATERN32, 1>(A, st);
    }
    return BASQ_EUNSUPPORTED;
}

// xs: exponential scheme (1: 2048-entry table + cubic, 1e-17; 2: 4096-entry table + quadratic, 2.5e-14, one instruction less)
static int dispatch_blocksum(int kk, int fam, const BlocksumArgs& A, hipStream_t st, int xs) {
    switch (kk) {
        case 1: return dispatch_blocksum_fam<1>(fam, A, st, xs);
        case 2: return dispatch_blocksum_fam<2>(fam, A, st, xs);
        case 3: return dispatch_blocksum_fam<3>(fam, A, st, xs);
        case 4: return dispatch_blocksum_fam<4>(fam, A, st, xs);
        case 5: return dispatch_blocksum_fam<5>(fam, A, st, xs);
        case 6: return dispatch_blocksum_fam<6>(fam, A, st, xs);
        case 7: return dispatch_blocksum_fam<7>(fam, A, st, xs);
        case 8: return dispatch_blocksum_fam<8>(fam, A, st, xs);
        case 9: return dispatch_blocksum_fam<9>(fam, A, st, xs);
        case 10: return dispatch_blocksum_fam<10>(fam, A, st, xs);
    }
    return BASQ_EUNSUPPORTED;
}

// ------------------------------------------------------------------------------------------------
// Block sums of SQUARED posterior covariances: the one term of the WSABI-M kernel (BASQ/_wsabi.py:227-249) that is
// not linear in the kernel,
//     E[j][s] = sum_{p in set s} (mu_p / 2) * cov(nys_j, y_p)^2,      cov = s2 k(nys_j, y_p) - sum_o B[j][o] ko[o][p]
// (B = k(nys, Xobs) W, ko[o][p] = s2 k(Xobs_o, y_p); + the likelihood noise on entry [kappa][kappa] of every kernel
// block, BASQ/_gp.py:275-276).  Same tile ownership, chunks and XCD map as blocksum_kernel; the correction is a second
// MFMA chain over the n_obs observations whose operands stream from L2 (B^T rows: 16 consecutive Nystrom rows of one
// observation = one 128-byte line; ko rows: 16 consecutive candidates of one observation).  Two blocks share every
// B^T fragment (JT + 2 loads per 2 JT MFMAs), the next fragments are in flight while the current ones multiply.
// Nothing of size [m, candidates] is ever written: 2 m n_obs flops per pair are what remains (MFMA-bound).
// ------------------------------------------------------------------------------------------------
struct SqArgs {
    const double* bmatT;   // [4 ko][ldb]  B^T, rows >= n_obs and columns >= m zero
    const double* kobs;    // [4 ko][ldk]  ko, LOCAL candidate positions, rows >= n_obs zero
    long long ldb, ldk;
    int ko;                // ceil(n_obs / 4)
    double outputscale, noise;
};

#ifndef BASQ_SQ_PF
#define BASQ_SQ_PF 2            // prefetch depth (observation steps) of the squared-covariance block sums: 1 / 2 / 3 -> 78 / 66 / 78 ms per
                                // config-5m batch (profiles/r03_z_wsabim_prefetch_depth_ab.txt; the one-step form of round 2: 74)
#endif
template <int KK, int FAM, int JT>
__device__ __forceinline__ void sq_pair_accumulate(const double (&a)[JT][KK], const CandFrag<KK>& f0, const CandFrag<KK>& f1,
                                                   long long row0, long long row1, int kap0, int kap1, int jrow,
                                                   const double* __restrict__ ap, int g, const SqArgs& Q,
                                                   double (&acc)[JT][4], const ExpK& ek, const double* tab) {
    d4 E0[JT], E1[JT];
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) {
        E0[jt] = d4{0.0, 0.0, 0.0, 0.0};
        E1[jt] = d4{0.0, 0.0, 0.0, 0.0};
    }
    const double* bp0 = Q.kobs + (long long)g * Q.ldk + row0;
    const double* bp1 = Q.kobs + (long long)g * Q.ldk + row1;
    const long long sa = 4 * Q.ldb, sb = 4 * Q.ldk;
    // Software pipeline, BASQ_SQ_PF observation steps deep: the fragments of step ko + PF are requested while step ko
    // multiplies (JT + 2 loads feed 2 JT matrix instructions per step; the loads come from L2 -- B^T rows -- and, for the
    // observation Gram block, from HBM).  Slots are indexed statically (the loop is unrolled PF-fold); steps past the end
    // re-read the last step's fragments.
    constexpr int PF = BASQ_SQ_PF;
    double av[PF][JT], b0[PF], b1[PF];
    const long long klast = Q.ko - 1;
#pragma unroll
    for (int p = 0; p < PF; ++p) {
        const long long kp_ = (p < Q.ko) ? p : klast;
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) av[p][jt] = ap[kp_ * sa + jt * 16];
        b0[p] = bp0[kp_ * sb];
        b1[p] = bp1[kp_ * sb];
    }
    for (int ko = 0; ko < Q.ko; ko += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            if (ko + p < Q.ko) {                                           // wave-uniform
                double avc[JT];
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) avc[jt] = av[p][jt];
                const double b0c = b0[p], b1c = b1[p];
                const long long kn = (ko + p + PF < Q.ko) ? (ko + p + PF) : klast;
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) av[p][jt] = ap[kn * sa + jt * 16];
                b0[p] = bp0[kn * sb];
                b1[p] = bp1[kn * sb];
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) {
                    E0[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(avc[jt], b0c, E0[jt], 0, 0, 0);
                    E1[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(avc[jt], b1c, E1[jt], 0, 0, 0);
                }
            }
        }
    }
    const bool noisy = Q.noise != 0.0;
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) {
        d4 D0 = {0.0, 0.0, 0.0, 0.0}, D1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            D0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[jt][kk], f0.b[kk], D0, 0, 0, 0);
            D1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[jt][kk], f1.b[kk], D1, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = jrow + jt * 16 + 4 * r;
            double v0 = __builtin_fma(Q.outputscale, kernel_from_arg_k<FAM, 1>(D0[r], ek, tab), -E0[jt][r]);
            double v1 = __builtin_fma(Q.outputscale, kernel_from_arg_k<FAM, 1>(D1[r], ek, tab), -E1[jt][r]);
            if (noisy) {
                if (j == kap0) v0 += Q.noise;
                if (j == kap1) v1 += Q.noise;
            }
            acc[jt][r] = __builtin_fma(f0.w * v0, v0, acc[jt][r]);
            acc[jt][r] = __builtin_fma(f1.w * v1, v1, acc[jt][r]);
        }
    }
}

template <int KK, int FAM, int JT>
__global__ void __launch_bounds__(256) blocksum_sq_kernel(const BlocksumArgs A, const SqArgs Q) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    constexpr int KP = KK * 4;
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;            // XCD-aware map: see blocksum_kernel
    const int jg = seq % A.n_jgroups;
    const int gidx = (seq / A.n_jgroups) * 8 + xcd;
    const int st = gidx % A.n_stiles;
    const int chunk = gidx / A.n_stiles;
    const int j0 = (jg * 4 + wave) * (16 * JT);
    __shared__ double exp_tab[ExpScheme<1>::N];
    exp_table_init<1>(exp_tab);
    if (chunk >= A.n_chunks) return;
    if (j0 >= A.m) return;
    const int s0 = st * 16;
    ExpK ek;
    expk_init<1>(ek);
    double a[JT][KK];
#pragma unroll
    for (int jt = 0; jt < JT; ++jt)
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) a[jt][kk] = A.nys[(long long)(j0 + jt * 16 + c) * KP + kk * 4 + g];
    double acc[JT][4];
#pragma unroll
    for (int jt = 0; jt < JT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[jt][r] = 0.0;
    const double* ap = Q.bmatT + (long long)g * Q.ldb + j0 + c;
    const int jrow = j0 + g;
    const bool col_ok = (s0 + c) < A.S;
    const long long lim = (A.off + A.Rl < A.n_full) ? (A.off + A.Rl) : A.n_full;
    // blocks of this chunk: a contiguous range, or -- residue-class mode, as in blocksum_kernel -- every class_mod-th block
    long long bA, bB, bstep;
    if (A.class_mod > 0) {
        bstep = A.class_mod;
        const long long cls = A.class0 + chunk;
        bA = A.blk_lo + ((cls - A.blk_lo % bstep) % bstep + bstep) % bstep;
        bB = A.blk_hi;
    } else {
        bstep = 1;
        bA = A.blk_lo + (long long)chunk * A.blk_per_chunk;
        bB = bA + A.blk_per_chunk;
        if (bB > A.blk_hi) bB = A.blk_hi;
    }
    // candidate of this lane's column in block i (weight mu / 2, zero when the position is not held here)
    auto frag = [&](CandFrag<KK>& f, long long pl, bool ok) -> long long {
        load_cand<KK>(f, A, pl, ok, g);
        f.w = 0.5 * f.wm;
        return ok ? pl : 0;
    };
    for (long long i = bA; i < bB; i += 2 * bstep) {
        const long long pg0 = i * A.S + s0 + c, pg1 = pg0 + bstep * A.S;
        CandFrag<KK> f0, f1;
        const long long r0 = frag(f0, pg0 - A.off, col_ok && pg0 >= A.off && pg0 < lim);
        const long long r1 = frag(f1, pg1 - A.off, col_ok && (i + bstep < bB) && pg1 >= A.off && pg1 < lim);
        // the noise sits on Nystrom row kappa = position inside the block = set index of the column
        sq_pair_accumulate<KK, FAM, JT>(a, f0, f1, r0, r1, s0 + c, s0 + c, jrow, ap, g, Q, acc, ek, exp_tab);
    }
    // ragged tail: all of it belongs to set S-1; tail point k meets the noise on Nystrom row k
    const long long t0 = (A.n_full > A.off) ? (A.n_full - A.off) : 0;
    if (chunk == A.n_chunks - 1 && st == A.n_stiles - 1 && t0 < A.Rl) {
        double tacc[JT][4];
#pragma unroll
        for (int jt = 0; jt < JT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) tacc[jt][r] = 0.0;
        for (long long p = t0; p < A.Rl; p += 32) {
            const long long p0 = p + c, p1 = p + 16 + c;
            CandFrag<KK> f0, f1;
            const long long r0 = frag(f0, p0, p0 < A.Rl);
            const long long r1 = frag(f1, p1, p1 < A.Rl);
            const long long k0 = A.off + p0 - A.n_full, k1 = A.off + p1 - A.n_full;
            sq_pair_accumulate<KK, FAM, JT>(a, f0, f1, r0, r1, (k0 < 0x7fffffffLL) ? (int)k0 : -1,
                                            (k1 < 0x7fffffffLL) ? (int)k1 : -1, jrow, ap, g, Q, tacc, ek, exp_tab);
        }
        const int c_last = (A.S - 1) - s0;
#pragma unroll
        for (int jt = 0; jt < JT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double v = sum16(tacc[jt][r]);
                if (c == c_last) acc[jt][r] += v;
            }
    }
    if (col_ok) {
        double* out = A.Xpart + (long long)chunk * A.m * A.S;
#pragma unroll
        for (int jt = 0; jt < JT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = j0 + jt * 16 + g + 4 * r;
                if (j < A.m) out[(long long)j * A.S + s0 + c] = acc[jt][r];
            }
    }
}

template <int KK, int FAM, int JT>
static int launch_blocksum_sq_jt(const BlocksumArgs& A, const SqArgs& Q, hipStream_t st) {
    BlocksumArgs B = A;
    B.n_jgroups = (A.m + 64 * JT - 1) / (64 * JT);
    const long long npairs = (long long)A.n_stiles * A.n_chunks;
    const long long nblk = ((npairs + 7) / 8) * 8 * B.n_jgroups;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return BASQ_EINVAL;
    hipLaunchKernelGGL((blocksum_sq_kernel<KK, FAM, JT>), dim3((unsigned)nblk), dim3(256), 0, st, B, Q);
    BASQ_CHECK_LAUNCH();
    return BASQ_OK;
}

template <int KK, int FAM>
static int launch_blocksum_sq(const BlocksumArgs& A, const SqArgs& Q, hipStream_t st) {
    static const int jt_env = [] { const char* e = getenv("BASQ_SQ_JT"); return e ? atoi(e) : 0; }();   // A/B knob
    if (BASQ_JT_FOR(KK) == 4 && jt_env != 2) return launch_blocksum_sq_jt<KK, FAM, 4>(A, Q, st);
    return launch_blocksum_sq_jt<KK, FAM, 2>(A, Q, st);
}

template <int KK>
static int dispatch_blocksum_sq_fam(int fam, const BlocksumArgs& A, const SqArgs& Q, hipStream_t st) {
    switch (fam) {
        case BASQ_FAMILY_RBF: return launch_blocksum_sq<KK, BASQ_FAMILY_RBF>(A, Q, st);
        case BASQ_FAMILY_MATERN52: return launch_blocksum_sq<KK, BASQ_FAMILY_MATERN52>(A, Q, st);
        case BASQ_FAMILY_MATERN32: return launch_blocksum_sq<KK, BASQ_FAMILY_MATERN32>(A, Q, st);
    }
    return BASQ_EUNSUPPORTED;
}

static int dispatch_blocksum_sq(int kk, int fam, const BlocksumArgs& A, const SqArgs& Q, hipStream_t st) {
    switch (kk) {
        case 1: return dispatch_blocksum_sq_fam<1>(fam, A, Q, st);
        case 2: return dispatch_blocksum_sq_fam<2>(fam, A, Q, st);
        case 3: return dispatch_blocksum_sq_fam<3>(fam, A, Q, st);
        case 4: return dispatch_blocksum_sq_fam<4>(fam, A, Q, st);
        case 5: return dispatch_blocksum_sq_fam<5>(fam, A, Q, st);
        case 6: return dispatch_blocksum_sq_fam<6>(fam, A, Q, st);
        case 7: return dispatch_blocksum_sq_fam<7>(fam, A, Q, st);
        case 8: return dispatch_blocksum_sq_fam<8>(fam, A, Q, st);
        case 9: return dispatch_blocksum_sq_fam<9>(fam, A, Q, st);
        case 10: return dispatch_blocksum_sq_fam<10>(fam, A, Q, st);
    }
    return BASQ_EUNSUPPORTED;
}

#ifndef BASQ_CAR_THREADS
#define BASQ_CAR_THREADS 1024
#endif
#ifndef BASQ_CHOL_THREADS
#define BASQ_CHOL_THREADS 1024   // work-group size of chol_inv_lds_kernel (multiple of 128)
#endif

// ------------------------------------------------------------------------------------------------
// Dense kernel matrix: wave = 64 rows x (CT x 16) columns, A fragments resident.
// ------------------------------------------------------------------------------------------------
template <int KK, int FAM>
__global__ void __launch_bounds__(256) gram_kernel(const double* __restrict__ pa, long long na,
                                                   const double* __restrict__ pb, long long nb, double scale,
                                                   double* __restrict__ K, long long ldk, int ctiles_per_block) {
    constexpr int KP = KK * 4, JT = 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const long long i0 = ((long long)blockIdx.y * 4 + wave) * 64;
    if (i0 >= na) return;
    double a[JT][KK];
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) {
        long long row = i0 + jt * 16 + c;
        if (row >= na) row = na - 1;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) a[jt][kk] = pa[row * KP + kk * 4 + g];
    }
    const long long ct0 = (long long)blockIdx.x * ctiles_per_block;
    for (int t = 0; t < ctiles_per_block; ++t) {
        const long long j0 = (ct0 + t) * 16;
        if (j0 >= nb) break;
        long long col = j0 + c;
        const bool ok = col < nb;
        if (!ok) col = nb - 1;
        double b[KK];
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) b[kk] = pb[col * KP + kk * 4 + g];
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
            d4 D = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) D = __builtin_amdgcn_mfma_f64_16x16x4f64(a[jt][kk], b[kk], D, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long long row = i0 + jt * 16 + g + 4 * r;
                if (ok && row < na) K[row * ldk + j0 + c] = scale * kernel_from_arg<FAM>(D[r]);
            }
        }
    }
}

template <int KK>
static int dispatch_gram_fam(int fam, const double* pa, long long na, const double* pb, long long nb, double scale,
                             double* K, long long ldk, hipStream_t st) {
    const int ct = 8;
    const long long ctiles = (nb + 15) / 16;
    dim3 grid((unsigned)((ctiles + ct - 1) / ct), (unsigned)((na + 255) / 256));
    switch (fam) {
        case BASQ_FAMILY_RBF:
            hipLaunchKernelGGL((gram_kernel<KK, BASQ_FAMILY_RBF>), grid, dim3(256), 0, st, pa, na, pb, nb, scale, K, ldk, ct);
            break;
        case BASQ_FAMILY_MATERN52:
            hipLaunchKernelGGL((gram_kernel<KK, BASQ_FAMILY_MATERN52>), grid, dim3(256), 0, st, pa, na, pb, nb, scale, K, ldk, ct);
            break;
        case BASQ_FAMILY_MATERN32:
            hipLaunchKernelGGL((gram_kernel<KK, BASQ_FAMILY_MATERN32>), grid, dim3(256), 0, st, pa, na, pb, nb, scale, K, ldk, ct);
            break;
        default: return BASQ_EUNSUPPORTED;
    }
    BASQ_CHECK_LAUNCH();
    return BASQ_OK;
}

// ------------------------------------------------------------------------------------------------
// f64 MFMA GEMM:  Cpart[z][M,N] = sum_{k in slice z} A[M,K] * (sum_c B[c][K,N]).  A[r][k] is read at r * lda + k * a_ks:
// (lda, 1) for a row-major A, (1, ld) for a TRANSPOSED copy -- the form the projections use: the 16 lanes of an MFMA
// row group then read 16 consecutive rows of one k (one 128-byte line) instead of 16 lines 8*lda bytes apart.
// Wave tile (16 JT) x 16.  Used for the Nystrom-feature contraction (BASQ/_rchq.py:88) and as the
// generic GEMM of the randomised SVD.
// ------------------------------------------------------------------------------------------------
template <int JT>
__global__ void __launch_bounds__(256) gemm_kernel(const double* __restrict__ A, long long lda, long long a_ks,
                                                   const double* __restrict__ B, long long ldb, long long bstride,
                                                   int nsum, double* __restrict__ C, long long ldc, long long cstride,
                                                   int M, int N, int K, int kslice, double alpha, int zdiv,
                                                   long long b_zstride) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int r0 = blockIdx.x * (16 * JT);
    const int n0 = (blockIdx.y * 4 + wave) * 16;
    if (n0 >= N) return;
    // blockIdx.z = (B slab) * zdiv + (K slice): one launch covers every chunk partial of the projection
    B += (long long)(blockIdx.z / zdiv) * b_zstride;
    const int k0 = (blockIdx.z % zdiv) * kslice;
    int k1 = k0 + kslice;
    if (k1 > K) k1 = K;
    const int col = (n0 + c < N) ? (n0 + c) : (N - 1);
    long long arow[JT];
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) {
        int r = r0 + jt * 16 + c;
        if (r >= M) r = M - 1;
        arow[jt] = (long long)r * lda;
    }
    d4 acc[JT];
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) acc[jt] = (d4){0.0, 0.0, 0.0, 0.0};
    // four k-steps per trip: all A/B loads of the trip are issued before its MFMAs
    for (int k = k0; k < k1; k += 16) {
        double bv[4], av[4][JT];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int kg = k + 4 * u + g;
            const bool kin = kg < k1;
            double b = 0.0;
            if (kin) {
                const double* bp = B + (long long)kg * ldb + col;
                b = bp[0];
                for (int v = 1; v < nsum; ++v) b += bp[(long long)v * bstride];
            }
            bv[u] = b;
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) av[u][jt] = kin ? A[arow[jt] + kg * a_ks] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int jt = 0; jt < JT; ++jt)
                acc[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][jt], bv[u], acc[jt], 0, 0, 0);
    }
    if (n0 + c < N) {
        double* Cz = C + (long long)blockIdx.z * cstride;
#pragma unroll
        for (int jt = 0; jt < JT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = r0 + jt * 16 + g + 4 * r;
                if (row < M) Cz[(long long)row * ldc + n0 + c] = alpha * acc[jt][r];
            }
    }
}

// ------------------------------------------------------------------------------------------------
// Tall-skinny f64 GEMM for the range finder (torch.svd_lowrank, BASQ/_rchq.py:29: A Q, A^T Q, Q^T A, X^T X with
// A the [m, m] Nystrom Gram matrix and only q + 1 <= 208 columns on the other side):
//     Cpart[z][M, N] = sum over K slice z of  op(A)[M, K] @ B[K, N],   op(A) = A ([M, K] row-major) or A^T (A is [K, M]).
// A wave owns 16 JT rows and ALL N columns (NT tiles of 16): its A fragments are read from HBM exactly once per launch.
// A row-major: lane (c, g) reads A[row c][k + 4 g .. + 3] -- one 128-byte line per row and 16-k trip -- and the k index
// of step u is k + 4 g + u (any order of the contraction index is a valid order, B is read to match).  Split K over the
// grid; the slabs are added in slice order by sum_parts_kernel (fixed summation order).
//
// The products run on v_mfma_f64_4x4x4_4b_f64 (four independent 4x4x4 products per instruction; lane map found by
// one-hot probing, tools/microbench_mfma4.hip: A lane = 16 k + 4 b + i, B lane = 16 k + 4 b + j, D lane = 16 i + 4 b + j).
// On gfx950 both fp64 matrix instructions run on the vector fp64 pipe: this one issues once per 16.5 cycles = 4 wave-wide
// FMAs' worth of pipe time for 256 lane-FMAs (75.7 TF/s chip-wide), v_mfma_f64_16x16x4 once per ~106 cycles for 1024
// (47.4 TF/s) -- profiles/r02_l_microbench_mfma_f64_4x4x4.txt.  The price is operand traffic: four times as many operand
// registers per flop.  So the fragments are LOADED exactly as for the 16x16x4 form (lane (c, g): row / column c,
// contraction index by g) and the four products of a 16 x 16 tile come from four ROTATIONS of the A fragment inside each
// row of 16 lanes (DPP row_ror 0/4/8/12: two v_mov per rotation, off the fp64 pipe): with rotation rho, block b = c >> 2
// multiplies the rows of lane group rg(rho, b) with its own four columns, i.e. D of lane (g, c) is
// C[16 jt + 4 rg + g][16 nt + c].  rg is read back from the same DPP applied to the lane index, so the code does not
// depend on the direction of the rotation.
// ------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ double dpp_row_f64(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_mov_dpp((int)b, CTRL, 0xf, 0xf, true);       // a rotation writes every lane: no "old" value
    const int hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// NT full column tiles of 16 + REM column GROUPS of 4 (N <= 16 NT + 4 REM).  A group costs ONE product per A fragment instead
// of four: its B fragment holds the group's 4 columns in all four blocks and the A fragment goes in un-rotated, so block b
// multiplies rows 4 b .. 4 b + 3 with those columns -- D of lane (g, c) is C[16 jt + 4 (c >> 2) + g][16 NT + 4 r + (c & 3)].
// At N = 99 that is 25 products per fragment and step instead of 28.
template <int NT, int REM, int JT, bool TRANS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2)))
skinny_gemm_kernel(const double* __restrict__ A, long long lda, long long a_bstride, const double* __restrict__ B,
                   long long ldb, double* __restrict__ C, long long ldc, long long cstride, int M, int N, int K, int kslice,
                   int nz) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    // blockIdx.y = member of a batch of products that share B (the projection of every chunk partial): its own A, its own
    // nz slabs of C
    A += (long long)blockIdx.y * a_bstride;
    C += (long long)blockIdx.y * nz * cstride;
    // Work-groups are dealt round-robin over the 8 XCDs (blockIdx.x % 8): with the K slice = blockIdx.x % nz and nz a
    // multiple of 8, an XCD only ever reads ITS slices of B, which then stay in its 4-MB L2 (B as a whole does not fit).
    const int zslice = blockIdx.x % nz;
    const int r0 = ((blockIdx.x / nz) * 4 + wave) * (16 * JT);   // may lie past M: such a wave still helps staging B, stores nothing
    const int k0 = zslice * kslice;
    int k1 = k0 + kslice;
    if (k1 > K) k1 = K;
    int rg[4];   // row group (of 4 rows) this lane's block multiplies under rotation rho
    rg[0] = c >> 2;
    rg[1] = (__builtin_amdgcn_update_dpp(0, c, 0x124, 0xf, 0xf, false) & 15) >> 2;
    rg[2] = (__builtin_amdgcn_update_dpp(0, c, 0x128, 0xf, 0xf, false) & 15) >> 2;
    rg[3] = (__builtin_amdgcn_update_dpp(0, c, 0x12c, 0xf, 0xf, false) & 15) >> 2;
    long long aoff[JT];
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) {
        int r = r0 + jt * 16 + c;
        if (r >= M) r = M - 1;
        aoff[jt] = TRANS ? (long long)r : (long long)r * lda;
    }
    constexpr int NB = NT + REM;                 // B fragments per step: NT tiles + REM column groups
    constexpr int W = 16 * NT + 4 * REM;         // columns of B the kernel works on
    int bcol[NB];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bcol[nt] = (nt * 16 + c < N) ? (nt * 16 + c) : (N - 1);
#pragma unroll
    for (int r = 0; r < REM; ++r) bcol[NT + r] = (16 * NT + 4 * r + (c & 3) < N) ? (16 * NT + 4 * r + (c & 3)) : (N - 1);
    double acc[JT][4][NT], accr[JT][REM > 0 ? REM : 1];
#pragma unroll
    for (int jt = 0; jt < JT; ++jt)
#pragma unroll
        for (int r = 0; r < (REM > 0 ? REM : 1); ++r) accr[jt][r] = 0.0;
#pragma unroll
    for (int jt = 0; jt < JT; ++jt)
#pragma unroll
        for (int rho = 0; rho < 4; ++rho)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[jt][rho][nt] = 0.0;
    auto kidx = [&](int k, int u) { return TRANS ? (k + 4 * u + g) : (k + 4 * g + u); };
    const bool a_vec = !TRANS && ((lda & 1) == 0) && (((uintptr_t)A & 15) == 0);
    auto mfma_step = [&](const double (&a)[JT], const double (&bv)[NB]) {
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
            double ar[4];
            ar[0] = a[jt];
            ar[1] = dpp_row_f64<0x124>(ar[0]);
            ar[2] = dpp_row_f64<0x128>(ar[0]);
            ar[3] = dpp_row_f64<0x12c>(ar[0]);
#pragma unroll
            for (int rho = 0; rho < 4; ++rho)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[jt][rho][nt] = __builtin_amdgcn_mfma_f64_4x4x4f64(ar[rho], bv[nt], acc[jt][rho][nt], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < REM; ++r)
                accr[jt][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[jt], bv[NT + r], accr[jt][r], 0, 0, 0);
        }
    };

    // Full 16-k trips.  The four waves of a work-group share the K slice, hence B: its [16, 16 NT] tile of a trip goes
    // through LDS (each thread fetches NT doubles, once per work-group instead of once per wave -- measured, the B
    // fragments through the vector memory path cost as much time as streaming A from HBM, and the two do not overlap:
    // profiles/r02_l_skinny_gemm_operand_paths.txt), double-buffered with one barrier per trip; the A fragments of the
    // NEXT trip (HBM latency) are requested a trip ahead.  Every address is a wave-uniform pointer that advances on the
    // scalar unit plus one per-lane offset: the matrix instructions leave no idle issue slots to hide index arithmetic in.
    // The staging reads are 16 NT columns wide whatever N is: columns >= N of a row are the head of the following row(s)
    // -- finite or not, they only reach accumulator columns that are never stored -- so the pipelined trips stop short of
    // the last over_rows rows of B, where such a read would leave the matrix.
    constexpr int NLD = (W + 15) / 16;           // staging loads per thread (the last one partial when REM > 0)
    __shared__ double btile[2][16][W];
    const int over_rows = (W + (int)ldb - 1) / (int)ldb;
    const int kfast_end = (k1 < K - over_rows) ? k1 : (K - over_rows);
    const int ktrips = (kfast_end > k0) ? (kfast_end - k0) / 16 : 0;
    int kdone = k0;
    if (ktrips > 0 && (TRANS || a_vec)) {                         // work-group uniform
        const int skk = threadIdx.x >> 4, sc = threadIdx.x & 15;   // staging: thread -> (row of the tile, column in a tile)
        const char* bbase = reinterpret_cast<const char*>(B + (long long)k0 * ldb);
        const unsigned soff = (unsigned)(((long long)skk * ldb + sc) * 8);
        const long long btrip = 16 * ldb * 8;
        auto stage_load = [&](double (&sr)[NLD]) {
#pragma unroll
            for (int i = 0; i < NLD; ++i)
                if (16 * i + 16 <= W || sc + 16 * i < W) sr[i] = *reinterpret_cast<const double*>(bbase + soff + 128 * i);
        };
        auto stage_store = [&](int buf, const double (&sr)[NLD]) {
#pragma unroll
            for (int i = 0; i < NLD; ++i)
                if (16 * i + 16 <= W || sc + 16 * i < W) btile[buf][skk][sc + 16 * i] = sr[i];
        };
        auto lds_b = [&](int buf, int u, double (&bv)[NB]) {
            const int kk = TRANS ? (4 * u + g) : (4 * g + u);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bv[nt] = btile[buf][kk][16 * nt + c];
#pragma unroll
            for (int r = 0; r < REM; ++r) bv[NT + r] = btile[buf][kk][16 * NT + 4 * r + (c & 3)];
        };
        const double* ap[JT];
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) ap[jt] = TRANS ? (A + (long long)(k0 + g) * lda + aoff[jt]) : (A + aoff[jt] + k0 + 4 * g);
        const long long atrip = TRANS ? 16 * lda : 16;
        auto fetch_a = [&](double (&av)[4][JT]) {
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) {
                if (TRANS) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) av[u][jt] = ap[jt][(long long)(4 * u) * lda];
                } else {
                    const double2* p = reinterpret_cast<const double2*>(ap[jt]);
                    const double2 lo = p[0], hi = p[1];
                    av[0][jt] = lo.x;
                    av[1][jt] = lo.y;
                    av[2][jt] = hi.x;
                    av[3][jt] = hi.y;
                }
            }
        };
        double av[4][JT], bv[NB], sr[NLD];
        fetch_a(av);
        stage_load(sr);
        stage_store(0, sr);
        __syncthreads();
        lds_b(0, 0, bv);
        for (int t = 0; t < ktrips; ++t) {
            const bool last = (t + 1 == ktrips);
            const int buf = t & 1;
            if (!last) {
                bbase += btrip;
                stage_load(sr);
            }
            const long long ainc = last ? 0 : atrip;
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) ap[jt] += ainc;
            double avn[4][JT];
            fetch_a(avn);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                double bvn[NB];
                if (u < 3) lds_b(buf, u + 1, bvn);
                mfma_step(av[u], bv);
                if (u < 3) {
#pragma unroll
                    for (int nt = 0; nt < NB; ++nt) bv[nt] = bvn[nt];
                }
            }
            if (!last) stage_store(buf ^ 1, sr);
            // one barrier per trip: everybody has finished READING tile buf (its fragments were consumed above) and WRITING
            // tile buf ^ 1; the next trip reads buf ^ 1 and overwrites buf
            __syncthreads();
            if (!last) lds_b(buf ^ 1, 0, bv);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) av[u][jt] = avn[u][jt];
        }
        kdone = k0 + 16 * ktrips;
    }
    // What remains (a partial last trip, or a row-major A that is not 16-byte aligned): masked loads, no pipeline.
    for (int k = kdone; k < k1; k += 16) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int kk = kidx(k, u);
            const int kc = (kk < k1) ? kk : (k1 - 1);
            double a[JT], bv[NB];
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) {
                const double v = TRANS ? A[(long long)kc * lda + aoff[jt]] : A[aoff[jt] + kc];
                a[jt] = (kk < k1) ? v : 0.0;
            }
            const double* bp = B + (long long)kc * ldb;
#pragma unroll
            for (int nt = 0; nt < NB; ++nt) bv[nt] = bp[bcol[nt]];
            mfma_step(a, bv);
        }
    }
    double* Cz = C + (long long)zslice * cstride;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        if (nt * 16 + c >= N) continue;
#pragma unroll
        for (int jt = 0; jt < JT; ++jt)
#pragma unroll
            for (int rho = 0; rho < 4; ++rho) {
                const int row = r0 + jt * 16 + 4 * rg[rho] + g;
                if (row < M) Cz[(long long)row * ldc + nt * 16 + c] = acc[jt][rho][nt];
            }
    }
#pragma unroll
    for (int r = 0; r < REM; ++r) {
        const int col = 16 * NT + 4 * r + (c & 3);
        if (col >= N) continue;
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
            const int row = r0 + jt * 16 + 4 * (c >> 2) + g;
            if (row < M) Cz[(long long)row * ldc + col] = accr[jt][r];
        }
    }
}

template <int NT, int REM, int JT>
static void launch_skinny(bool trans, int M, int nz, int nbatch, hipStream_t st, const double* A, long long lda,
                          long long a_bstride, const double* B, long long ldb, double* C, long long ldc, long long cstride,
                          int N, int K, int kslice) {
    dim3 grid((unsigned)((M + 64 * JT - 1) / (64 * JT)) * (unsigned)nz, (unsigned)nbatch);
    if (trans)
        hipLaunchKernelGGL((skinny_gemm_kernel<NT, REM, JT, true>), grid, dim3(256), 0, st, A, lda, a_bstride, B, ldb, C, ldc,
                           cstride, M, N, K, kslice, nz);
    else
        hipLaunchKernelGGL((skinny_gemm_kernel<NT, REM, JT, false>), grid, dim3(256), 0, st, A, lda, a_bstride, B, ldb, C, ldc,
                           cstride, M, N, K, kslice, nz);
}

// rows per wave: 32 while the accumulators (4 NT JT doubles) leave room for two waves per SIMD, 16 for wide outputs
static void dispatch_skinny(bool trans, int M, int nz, int nbatch, hipStream_t st, const double* A, long long lda,
                            long long a_bstride, const double* B, long long ldb, double* C, long long cstride, int N, int K,
                            int kslice) {
    const int nt = (N + 15) / 16, ng = (N + 3) / 4;
    // Small launches (the Gram products of the orthonormalisations, the per-round projections of the few candidates the
    // class messages do not cover) are latency chains of 16-k trips: 16 rows per wave halve the products per trip, and the
    // twice as many waves still fit the chip several times over.
    const bool small = (long long)((M + 127) / 128) * nz * nbatch <= 128;
    if (small && ng == 25) launch_skinny<6, 1, 1>(trans, M, nz, nbatch, st, A, lda, a_bstride, B, ldb, C, N, cstride, N, K, kslice);
    else if (small && nt <= 4) launch_skinny<4, 0, 1>(trans, M, nz, nbatch, st, A, lda, a_bstride, B, ldb, C, N, cstride, N, K, kslice);
    else if (small && nt <= 7) launch_skinny<7, 0, 1>(trans, M, nz, nbatch, st, A, lda, a_bstride, B, ldb, C, N, cstride, N, K, kslice);
    // the two batch sizes of the BASELINE configurations (q = 99, 199) get their exact column-group count
    else if (ng == 25) launch_skinny<6, 1, 2>(trans, M, nz, nbatch, st, A, lda, a_bstride, B, ldb, C, N, cstride, N, K, kslice);
    else if (ng == 50) launch_skinny<12, 2, 1>(trans, M, nz, nbatch, st, A, lda, a_bstride, B, ldb, C, N, cstride, N, K, kslice);
    else if (nt <= 4) launch_skinny<4, 0, 2>(trans, M, nz, nbatch, st, A, lda, a_bstride, B, ldb, C, N, cstride, N, K, kslice);
    else if (nt <= 7) launch_skinny<7, 0, 2>(trans, M, nz, nbatch, st, A, lda, a_bstride, B, ldb, C, N, cstride, N, K, kslice);
    else launch_skinny<13, 0, 1>(trans, M, nz, nbatch, st, A, lda, a_bstride, B, ldb, C, N, cstride, N, K, kslice);
}

// Xsum[e] = sum_c Xpart[c][e]  (chunk order): one streaming pass instead of one GEMM per chunk partial
__global__ void chunk_sum_kernel(const double* __restrict__ Xpart, long long n, int n_chunks, double* __restrict__ Xsum) {
    const long long e = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (e + 1 < n) {
        double2 v = *reinterpret_cast<const double2*>(Xpart + e);
        for (int c = 1; c < n_chunks; ++c) {
            const double2 u = *reinterpret_cast<const double2*>(Xpart + (long long)c * n + e);
            v.x += u.x;
            v.y += u.y;
        }
        *reinterpret_cast<double2*>(Xsum + e) = v;
    } else if (e < n) {
        double v = Xpart[e];
        for (int c = 1; c < n_chunks; ++c) v += Xpart[(long long)c * n + e];
        Xsum[e] = v;
    }
}

// out[0][s] = sum_c totpart[c][s];  out[1+r][s] = sum_z work[z][r][s]   (fixed order)
// sum_{z < n} p[z * stride] in index order, the loads issued eight at a time: a plain loop over a run-time count waits for
// every load before it issues the next one (one L2 / memory round trip per term -- 48 terms cost 13 us).
__device__ __forceinline__ double ordered_strided_sum(const double* __restrict__ p, int n, long long stride) {
    double v = 0.0;
    int z = 0;
    for (; z + 8 <= n; z += 8) {
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = p[(long long)(z + u) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) v += t[u];
    }
    for (; z < n; ++z) v += p[(long long)z * stride];
    return v;
}

__global__ void project_reduce_kernel(const double* __restrict__ work, int ksplit, int q, int S,
                                      const double* __restrict__ totpart, int n_chunks, double* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (q + 1) * S) return;
    const int r = idx / S, s = idx % S;
    double v = 0.0;
    if (r == 0) {
        for (int cc = 0; cc < n_chunks; ++cc) v += totpart[(long long)cc * S + s];
    } else {
        v = ordered_strided_sum(work + (long long)(r - 1) * S + s, ksplit, (long long)q * S);
    }
    out[idx] = v;
}

__global__ void finalize_kernel(const double* __restrict__ parts, int n_parts, int msg_rows, int q, int S,
                                const double* __restrict__ diagU, long long ld_diag, int n_diag, double diag_noise,
                                int diag_wrow, int diag_tail_row, int n_tail_diag, double* __restrict__ XcarT,
                                double* __restrict__ tot_out, const long long* __restrict__ geo) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (q + 1) * S) return;
    if (geo) {                                     // descriptor-driven round: n_tail_diag is a cap, the tail length is on the device
        const long long nt = geo[5];
        if (nt < n_tail_diag) n_tail_diag = (int)nt;
    }
    const int r = idx / S, s = idx % S;
    const long long stride = (long long)msg_rows * S;
    const double tot = ordered_strided_sum(parts + s, n_parts, stride);
    if (r == 0) {
        XcarT[idx] = 1.0;
        tot_out[s] = tot;
        return;
    }
    double v = ordered_strided_sum(parts + idx, n_parts, stride);
    if (diagU) {
        const bool tail_set = diag_tail_row != 0 && s == S - 1;
        const double* urow = diagU + (long long)(r - 1) * ld_diag;
        if (s < n_diag) {
            // entry [s][s] of every FULL block: weight of set s without the ragged tail (which is its own block)
            double wgt = tot;
            if (diag_wrow != 0) {
                wgt = 0.0;
                for (int p = 0; p < n_parts; ++p) wgt += parts[p * stride + (long long)diag_wrow * S + s];
            }
            if (tail_set) {
                double tw = 0.0;
                for (int k = 0; k < S; ++k)
                    for (int p = 0; p < n_parts; ++p) tw += parts[p * stride + (long long)diag_tail_row * S + k];
                wgt -= tw;
            }
            v += diag_noise * wgt * urow[s];
        }
        if (tail_set) {
            // entry [k][k] of the tail block: tail point k meets Nystrom row k, and the tail belongs to the last set
            double acc = 0.0;
            for (int k = 0; k < n_tail_diag; ++k) {
                double tw = 0.0;
                for (int p = 0; p < n_parts; ++p) tw += parts[p * stride + (long long)diag_tail_row * S + k];
                acc += tw * urow[k];
            }
            v += diag_noise * acc;
        }
    }
    XcarT[idx] = v / tot;
}

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

// ------------------------------------------------------------------------------------------------
// Caratheodory elimination (BASQ/_rchq.py:146-175), single work-group, reference op order.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) car_eliminate_kernel(double* __restrict__ PhiT, double* __restrict__ mu_g,
                                                             int M, int s, int* __restrict__ keep_rank,
                                                             int* __restrict__ kept, double* __restrict__ w_star,
                                                             int* __restrict__ info) {
#pragma clang fp contract(off)   // plain operators below: the reference rounds after every mul / sub / div
    // (HIP's __dmul_rn/__dsub_rn are inline functions compiled with contraction allowed: they DO fuse)
    __shared__ double mu[1024];
    __shared__ double pc[1024];
    __shared__ double red_v[16];
    __shared__ int red_i[16];
    __shared__ double sh_alpha, sh_phij;
    __shared__ int sh_j;
    __shared__ int wave_cnt[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nrows = M - s;
    const double INF = __builtin_huge_val();
    mu[tid] = (tid < M) ? mu_g[tid] : 0.0;
    int status = 0;
    // update-pass geometry: thread -> fixed column i, rows strided
    const int rows_per_pass = 1024 / M;
    const int my_i = tid % M, my_r = tid / M;
    const bool upd = my_r < rows_per_pass;
    __syncthreads();
    for (int k = 0; k < nrows; ++k) {
        const double* col = PhiT + (long long)k * M;
        const double phi = (tid < M) ? col[tid] : 0.0;
        const bool pos = (tid < M) && (phi > 0.0);
        double av = pos ? (mu[tid] / phi) : INF;
        int ai = pos ? tid : 0x7fffffff;
        // first-index argmin (torch.argmin semantics, :152)
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const double ov = __shfl_xor(av, o, 64);
            const int oi = __shfl_xor(ai, o, 64);
            if (ov < av || (ov == av && oi < ai)) { av = ov; ai = oi; }
        }
        if (lane == 0) { red_v[wv] = av; red_i[wv] = ai; }
        __syncthreads();
        if (wv == 0) {
            double v = (lane < 16) ? red_v[lane] : INF;
            int i = (lane < 16) ? red_i[lane] : 0x7fffffff;
#pragma unroll
            for (int o = 8; o >= 1; o >>= 1) {
                const double ov = __shfl_xor(v, o, 64);
                const int oi = __shfl_xor(i, o, 64);
                if (ov < v || (ov == v && oi < i)) { v = ov; i = oi; }
            }
            if (lane == 0) { sh_j = i; sh_alpha = v; }
        }
        __syncthreads();
        const int j = sh_j;
        if (j == 0x7fffffff) { status = 1; break; }   // uniform: no positive entry (reference raises)
        if (tid == j) sh_phij = phi;
        const double aj = sh_alpha;
        if (tid < M) {                                                                  // :158-159
            const double step = aj * phi;
            mu[tid] = (tid == j) ? 0.0 : (mu[tid] - step);
        }
        for (int cc = k + 1 + tid; cc < nrows; cc += 1024) pc[cc] = PhiT[(long long)cc * M + j];
        __syncthreads();
        const double phij = sh_phij;
        const double rphij = 1.0 / phij;                                     // correctly rounded reciprocal
        const double phi_i = col[my_i];
        if (upd) {
            for (int cc = k + 1 + my_r; cc < nrows; cc += rows_per_pass) {   // :165-171
                double* p = PhiT + (long long)cc * M + my_i;
                const double prod = pc[cc] * phi_i;
                const double o = div_by_recip(prod, phij, rphij);            // == prod / phij, bit for bit
                *p = (my_i == j) ? 0.0 : (*p - o);
            }
        }
        __syncthreads();
    }
    // survivors: mu > 0 (:173-174), ascending
    const bool keep = (tid < M) && (mu[tid] > 0.0);
    const unsigned long long bal = __ballot(keep);
    if (lane == 0) wave_cnt[wv] = __popcll(bal);
    __syncthreads();
    int base = 0, total = 0;
    for (int w = 0; w < 16; ++w) {
        if (w < wv) base += wave_cnt[w];
        total += wave_cnt[w];
    }
    const int rank = base + __popcll(bal & ((1ull << lane) - 1ull));
    if (tid < M) {
        keep_rank[tid] = keep ? rank : -1;
        if (keep) { kept[rank] = tid; w_star[rank] = mu[tid]; }     // (mu_g stays as it came: ABI 13)
    }
    if (tid == 0) { info[0] = total; info[1] = status; }
}

// LDS-resident form (used when (M-s)*M doubles fit in 160 KB, e.g. M = 200, s = 100): identical arithmetic and
// pivot rule, but the null-space rows never leave the CU and a step costs ONE barrier:
//   * ratio test: wave minimum by DPP (min is exact, so any association gives the reference's value), first
//     lane holding it by ballot; the per-wave winners (value, index, reciprocal of the pivot entry -- one IEEE
//     divide per step instead of one per thread) go through LDS and EVERY wave scans them;
//   * an eliminated column is remembered in a per-thread flag instead of being zeroed (:167-171 zero it only so
//     that it is never chosen again): Phi[:, j] is then read-only during the rank-1 update and needs no staging;
//   * software pipeline: thread (column i, row group 0) updates row k+1 FIRST and, holding the fresh entry and its
//     own weight in registers, runs the ratio test of step k+1 at once -- concurrently with the other waves'
//     updates of rows k+2.. -- so the test is off the critical path; weights never touch memory.
__device__ __forceinline__ double wave_min_f64(double v) {
    const double INF = __builtin_huge_val();
    v = fmin(v, dpp_shift_fill_f64<0x111, 0xf>(v, INF));      // row_shr 1, 2, 4, 8: running minima inside rows of 16
    v = fmin(v, dpp_shift_fill_f64<0x112, 0xf>(v, INF));
    v = fmin(v, dpp_shift_fill_f64<0x114, 0xf>(v, INF));
    v = fmin(v, dpp_shift_fill_f64<0x118, 0xf>(v, INF));
    v = fmin(v, dpp_shift_fill_f64<0x142, 0xa>(v, INF));      // row_bcast 15 / 31: lane 63 ends with the wave minimum
    v = fmin(v, dpp_shift_fill_f64<0x143, 0xc>(v, INF));
    return readlane_f64(v, 63);
}

__global__ void __launch_bounds__(1024) car_eliminate_lds_kernel(const double* __restrict__ PhiT_g,
                                                                 double* __restrict__ mu_g, int M, int s,
                                                                 int* __restrict__ keep_rank, int* __restrict__ kept,
                                                                 double* __restrict__ w_star, int* __restrict__ info) {
#pragma clang fp contract(off)   // plain operators: the reference rounds after every mul / sub / div
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int nrows = M - s;
    double* Phi = sm;                          // [nrows][M]
    __shared__ double red_v[2][16];            // per-wave winners, double-buffered by step parity
    __shared__ double red_r[2][16];            // 1 / phi of each wave's winner (the pivot's reciprocal, computed once)
    __shared__ int red_i[2][16];
    __shared__ int wave_cnt[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nt = blockDim.x, nwv = nt >> 6;       // 256..1024 threads (BASQ_CAR_THREADS)
    const int nwv_act = (M + 63) >> 6;              // waves that own a column (tid < M)
    const double INF = __builtin_huge_val();
    for (int e = tid; e < nrows * M; e += nt) Phi[e] = PhiT_g[e];
    double mu_r = (tid < M) ? mu_g[tid] : 0.0;      // weight of column tid: only this thread ever touches it
    int status = 0;
    const int rows_per_pass = nt / M > 0 ? nt / M : 1;   // M <= nt is guaranteed by the launcher
    const int my_i = tid % M, my_r = tid / M;       // tid < M  <=>  my_r == 0 and my_i == tid
    const bool upd = my_r < rows_per_pass;
    bool dead = false;                              // column my_i has been eliminated
    __syncthreads();
    // ratio test on (phi = entry of the current null vector in column tid, mu_r): publishes this wave's winner
    auto ratio_test = [&](double phi, int parity) {
        const bool pos = (tid < M) && !dead && (phi > 0.0);
        const double av = pos ? (mu_r / phi) : INF;
        const double rphi = pos ? (1.0 / phi) : 0.0;     // second, independent divide: shares the latency of the first
        const double wmin = wave_min_f64(av);
        const unsigned long long hit = __ballot(pos && av == wmin);   // first-index argmin (torch.argmin, :152)
        const int first = hit ? (int)__builtin_ctzll(hit) : 0;
        if (lane == first) {
            red_v[parity][wv] = wmin;
            red_r[parity][wv] = rphi;
            red_i[parity][wv] = hit ? (wv * 64 + first) : 0x7fffffff;
        }
    };
    if (nrows > 0 && wv < nwv_act) ratio_test((tid < M) ? Phi[tid] : 0.0, 0);
    __syncthreads();
    for (int k = 0; k < nrows; ++k) {
        const double* col = Phi + (size_t)k * M;
        const int pb = k & 1;
        BASQ_NS_STAMP(k, 0);
        // every wave scans the (<= 16) per-wave winners: lower wave wins ties
        double aj = INF, rphij = 0.0;
        int j = 0x7fffffff;
        if (nwv_act <= 4) {                             // M <= 256: all loads in flight at once
            const double v0 = red_v[pb][0], v1 = red_v[pb][1], v2 = red_v[pb][2], v3 = red_v[pb][3];
            const double r0 = red_r[pb][0], r1 = red_r[pb][1], r2 = red_r[pb][2], r3 = red_r[pb][3];
            const int i0 = red_i[pb][0], i1 = red_i[pb][1], i2 = red_i[pb][2], i3 = red_i[pb][3];
            if (i0 != 0x7fffffff) { aj = v0; j = i0; rphij = r0; }
            if (nwv_act > 1 && i1 != 0x7fffffff && (v1 < aj || j == 0x7fffffff)) { aj = v1; j = i1; rphij = r1; }
            if (nwv_act > 2 && i2 != 0x7fffffff && (v2 < aj || j == 0x7fffffff)) { aj = v2; j = i2; rphij = r2; }
            if (nwv_act > 3 && i3 != 0x7fffffff && (v3 < aj || j == 0x7fffffff)) { aj = v3; j = i3; rphij = r3; }
        } else {
            for (int w = 0; w < nwv_act; ++w) {
                const double v = red_v[pb][w];
                const int i = red_i[pb][w];
                if (i != 0x7fffffff && (v < aj || j == 0x7fffffff)) { aj = v; j = i; rphij = red_r[pb][w]; }
            }
        }
        if (j == 0x7fffffff) { status = 1; break; }   // uniform: no positive entry (reference raises)
        const double phij = col[j];                    // rphij = RN(1 / phij), from the pivot's own lane
        const double phi_i = col[my_i];
        if (my_i == j) dead = true;
        if (tid < M) {                                                                  // :158-159
            const double step = aj * phi_i;
            mu_r = dead ? 0.0 : (mu_r - step);         // eliminated columns: the reference has Phi = 0, mu = 0
        }
        BASQ_NS_STAMP(k, 3);
        int cc = k + 1 + my_r;
        double fresh = 0.0;
        if (upd && !dead && cc < nrows) {              // first row of this thread: row k+1 for the column owners
            double* p = Phi + (size_t)cc * M;
            const double o = div_by_recip(p[j] * phi_i, phij, rphij);     // == prod / phij, bit for bit
            fresh = p[my_i] - o;
            p[my_i] = fresh;
        }
        cc += rows_per_pass;
        if (k + 1 < nrows && wv < nwv_act) ratio_test(fresh, pb ^ 1);    // step k+1's test, off the critical path
        if (upd && !dead) {
            // four independent rows per trip, all LDS reads before the writes (otherwise every row is its own round trip).
            // (Taking the column owners off this loop -- they also carry the serial part of a step -- made the kernel
            // SLOWER, 188 vs 169 us at 100 x 200: the update is bound by LDS bandwidth, 24 B per entry and step, not by
            // the serial part, and every thread's share counts.)
            for (; cc + 3 * rows_per_pass < nrows; cc += 4 * rows_per_pass) {   // :165-171
                double* p0 = Phi + (size_t)cc * M;
                double* p1 = p0 + (size_t)rows_per_pass * M;
                double* p2 = p1 + (size_t)rows_per_pass * M;
                double* p3 = p2 + (size_t)rows_per_pass * M;
                const double a0 = p0[j], a1 = p1[j], a2 = p2[j], a3 = p3[j];
                const double b0 = p0[my_i], b1 = p1[my_i], b2 = p2[my_i], b3 = p3[my_i];
                const double o0 = div_by_recip(a0 * phi_i, phij, rphij);
                const double o1 = div_by_recip(a1 * phi_i, phij, rphij);
                const double o2 = div_by_recip(a2 * phi_i, phij, rphij);
                const double o3 = div_by_recip(a3 * phi_i, phij, rphij);
                p0[my_i] = b0 - o0;
                p1[my_i] = b1 - o1;
                p2[my_i] = b2 - o2;
                p3[my_i] = b3 - o3;
            }
            for (; cc < nrows; cc += rows_per_pass) {
                double* p = Phi + (size_t)cc * M;
                const double o = div_by_recip(p[j] * phi_i, phij, rphij);
                p[my_i] = p[my_i] - o;
            }
        }
        BASQ_NS_STAMP(k, 4);
        __syncthreads();
        BASQ_NS_STAMP(k, 5);
    }
    const bool keep = (tid < M) && (mu_r > 0.0);
    const unsigned long long bal = __ballot(keep);
    if (lane == 0) wave_cnt[wv] = __popcll(bal);
    __syncthreads();
    int base = 0, total = 0;
    for (int w = 0; w < nwv; ++w) {
        if (w < wv) base += wave_cnt[w];
        total += wave_cnt[w];
    }
    const int rank = base + __popcll(bal & ((1ull << lane) - 1ull));
    if (tid < M) {
        keep_rank[tid] = keep ? rank : -1;
        if (keep) { kept[rank] = tid; w_star[rank] = mu_r; }        // (mu_g stays as it came: ABI 13)
    }
    if (tid == 0) { info[0] = total; info[1] = status; }
}

// ------------------------------------------------------------------------------------------------
// Cluster kernels for the two per-round reductions (null space, elimination).
//
// Both reductions are chains of s (resp. M - s) dependent steps over a [rows x M] matrix; what a step costs is its
// synchronisation, not its arithmetic.  Common layout: the matrix lives in REGISTERS, row r in wave (r % W), slot
// (r / W); lanes own column PAIRS (slot k of a lane holds column 2 lane + (k & 1) + 128 (k >> 1)); a work-group is
// BASQ_WPG = 8 waves (two per SIMD) and a cluster is NCU work-groups (W = 8 NCU waves) that exchange ONE message
// per step:
//   NCU = 1: through LDS (ring buffer + one counter word, no s_barrier in the elimination);
//   NCU > 1: as TAGGED GRANULES in global memory -- every double travels as one 16-byte store {tag, low word, tag, high
//            word}, tag = the step it belongs to, and a reader sweeps its granules (L1-bypassing sc1 loads) until both
//            tags of each match: the data is its own flag, so there is no drain, no flag store and no acquire, ONE trip
//            per step (round 2's flag + drain + gather form took three: 2.24 -> 0.71 ms per 200 x 400 null space).
//            The members sit on every 8th work-group of the grid, which the dispatcher deals to ONE XCD; they check
//            that at launch (cluster_shares_xcd) and then use PLAIN stores, which stay in that XCD's L2 -- otherwise
//            write-through (sc1) stores.  Placement changes speed, never results.  The granule words are zeroed before
//            every launch; spins are bounded: a member that is not resident -> status 2 -> the caller's retry on the
//            single-work-group kernels.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(1))) unsigned long long basq_gu64;
typedef __attribute__((address_space(1))) unsigned int basq_gu32;
#define BASQ_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
#define BASQ_PAIRCOL(k) (2 * lane + ((k) & 1) + 128 * ((k) >> 1))
#define BASQ_WPG 8                        // waves per work-group of the cluster kernels: two per SIMD, 256 VGPRs each -- the
                                          // 100 x 200 matrix is 56 doubles per lane and stays in directly addressable VGPRs
                                          // (4 waves of 512 registers would park half of it in AGPRs: two moves per use)
#define BASQ_SPIN_LIMIT (1u << 22)       // polls (with s_sleep) before a cluster kernel gives up: ~0.3 s
#define BASQ_ABORT_COUNT 0x40000000

// Tagged granules: a double handed to another work-group as ONE 16-byte write-through store {tag, low word, tag, high word}.
// Each 8-byte half carries the tag, so a reader that finds both tags equal to the epoch it waits for holds the value -- the
// data is its own flag: no drain, no flag store, no second round trip (MI355X_MICROARCH.md, visibility: data-tagged
// granules; 8-byte halves are the unit observed untorn).  Words are zeroed by the launcher before every launch; epochs
// count steps within the launch and are never 0.
typedef unsigned basq_v4u __attribute__((ext_vector_type(4)));
// Same-XCD clusters: a PLAIN store leaves the granule in the XCD's L2, where a sibling's L1-bypassing load finds it in a
// fraction of the time a write-through line takes to come back from the fabric (guide, "stores of each flavour").  Which
// XCD a work-group runs on is not ours to choose, so the cluster checks it (cluster_shares_xcd) and falls back to
// write-through stores when its members are spread; `local` is work-group uniform.
__device__ __forceinline__ void granule_store(__amdgpu_buffer_rsrc_t rs, unsigned idx, unsigned tag, double v, bool local) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const basq_v4u g = {tag, (unsigned)b, tag, (unsigned)(b >> 32)};
    if (local) __builtin_amdgcn_raw_buffer_store_b128(g, rs, (int)(idx * 16u), 0, 0);
    else __builtin_amdgcn_raw_buffer_store_b128(g, rs, (int)(idx * 16u), 0, 16);
}
__device__ __forceinline__ basq_v4u granule_load(__amdgpu_buffer_rsrc_t rs, unsigned idx) {
    return __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(idx * 16u), 0, 16);     // sc1: served by L2 / the fabric, never L1
}
__device__ __forceinline__ bool granule_ok(const basq_v4u g, unsigned tag) { return g.x == tag && g.z == tag; }
__device__ __forceinline__ double granule_value(const basq_v4u g) {
    return __longlong_as_double((long long)(((unsigned long long)g.w << 32) | (unsigned long long)g.y));
}
// One handshake per launch: every member publishes the XCD it runs on (agent-scope word, zeroed by the launcher) and reads
// the others'.  -> true iff all NCU members share one XCD (a member that never answers counts as elsewhere).
template <int NCU>
__device__ __forceinline__ bool cluster_shares_xcd(unsigned* words, int cu, int* verdict_l) {
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const unsigned mine = 0x100u | (__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xfu);   // HW_REG_XCC_ID[3:0]
        if (lane == 0) __hip_atomic_store((basq_gu32*)(words + cu), mine, BASQ_RLX_AGENT);
        unsigned f = mine, spins = 0;
        for (;;) {
            f = (lane < NCU) ? __hip_atomic_load((basq_gu32*)(words + lane), BASQ_RLX_AGENT) : mine;
            if (__all(f != 0u) || ++spins > (1u << 16)) break;
            __builtin_amdgcn_s_sleep(1);
        }
        const bool same = __all(f == mine);
        if (lane == 0) *verdict_l = same ? 1 : 0;
    }
    __syncthreads();
    return *verdict_l != 0;
}
#define BASQ_GRANULE_SPIN_LIMIT (1u << 19)   // sweeps (~1 us each, with s_sleep after the first few) before a cluster gives up

__device__ __forceinline__ int wave_min_i32(int v) {
    const int BIG = 0x7fffffff;
    v = min(v, __builtin_amdgcn_update_dpp(BIG, v, 0x111, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(BIG, v, 0x112, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(BIG, v, 0x114, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(BIG, v, 0x118, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(BIG, v, 0x142, 0xa, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(BIG, v, 0x143, 0xc, 0xf, false));
    return __builtin_amdgcn_readlane(v, 63);
}

// slot kj (wave-uniform) of a register row
// (the empty asm pins each element in a VGPR first: left alone, LLVM rewrites the select chain into ONE load with a
// selected address, which forces the whole register-resident matrix into scratch memory)
template <int NV>
__device__ __forceinline__ double pick_slot(const double (&r)[NV], int kj) {
    double v = r[0];
    asm("" : "+v"(v));
#pragma unroll
    for (int k = 1; k < NV; ++k) {
        double x = r[k];
        asm("" : "+v"(x));
        v = (kj == k) ? x : v;
    }
    return v;
}

// Monotone counter in LDS shared by the waves of a ONE-work-group cluster (clusters of several work-groups hand over tagged
// granules instead and need no counter).
__device__ __forceinline__ void counter_publish(int* cnt, int value, int lane) {
    if (lane == 0) __hip_atomic_store(cnt, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// wait until *cnt > k; returns the value seen (>= BASQ_ABORT_COUNT: another wave gave up)
__device__ __forceinline__ int counter_wait_gt(int* cnt, int k) {
    unsigned spins = 0;
    int c;
    for (;;) {
        c = __hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
        c = __builtin_amdgcn_readfirstlane(c);
        if (c > k) break;
        if (++spins > BASQ_SPIN_LIMIT) {                           // never in a healthy run: abort the whole work-group
            __hip_atomic_store(cnt, BASQ_ABORT_COUNT, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            c = BASQ_ABORT_COUNT;
            break;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");        // no instruction: keeps the payload loads below the poll
    return c;
}

// ------------------------------------------------------------------------------------------------
// Caratheodory elimination (BASQ/_rchq.py:146-175), cluster form: bit-identical to the reference's op order
// (mul, Markstein quotient, sub per entry; IEEE divides in the ratio test; first-index argmin).
//   * null vector c lives in the registers of wave c % W; every wave keeps its own copy of the weights;
//   * the owner of row k+1 updates THAT row first, runs the ratio test of step k+1 on it and publishes
//     {row, j, alpha, 1/phi_j, phi_j} in a ring slot -- then catches up with its other rows.  The other waves only
//     consume: the dependent chain of a step is publish -> read -> one row update -> ratio test, while the rank-1
//     updates of the (M-s-k) remaining rows run beside it on the other SIMDs / CUs.  No barrier;
//   * ring of 2 W slots: a slot is rewritten W+1 steps later at the earliest, by which time every wave (each owns
//     one row in any W consecutive steps, and publishing needs the previous pivot) has consumed it.
// ------------------------------------------------------------------------------------------------
template <int NV, int NR, int NCU>
__global__ void __launch_bounds__(BASQ_WPG * 64) car_eliminate_cluster_kernel(const double* __restrict__ PhiT_g,
                                                                    double* __restrict__ mu_g, int M, int s,
                                                                    int* __restrict__ keep_rank, int* __restrict__ kept,
                                                                    double* __restrict__ w_star, int* __restrict__ info,
                                                                    double* ws, int cluster_stride) {
#pragma clang fp contract(off)   // plain operators: the reference rounds after every mul / sub / div
    constexpr int WPG = BASQ_WPG, W = WPG * NCU, NC = NV * 64, D = 2 * W, SLOT = NC + 8;
    constexpr bool GLOBAL = NCU > 1;
    if (blockIdx.x % cluster_stride) return;       // cluster members share `blockIdx.x % 8`: one XCD under round-robin
    const int cu = blockIdx.x / cluster_stride;    // placement -- speed only, nothing depends on it
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int gw = cu * WPG + wv;
    const int nrows = M - s;
    const double INF = __builtin_huge_val();
    __shared__ __attribute__((aligned(16))) double ring_l[GLOBAL ? 2 : D * SLOT];
    __shared__ int count_l;
    double* ring = ring_l;                          // one work-group: ring + counter word in LDS
    int* count = &count_l;
    // clusters: the ring holds tagged granules (tag = step + 1; every word zeroed by the launcher) -- a consumer sweeps the
    // slot until all its tags match: no counter, no drain on the publishing side, one fabric trip per step
    __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc((void*)(ws + 16), 0, GLOBAL ? (int)(D * SLOT * 16) : 0, 0x00020000);
    __shared__ int local_l;
    bool local = false;
    if (GLOBAL) local = cluster_shares_xcd<NCU>((unsigned*)ws, cu, &local_l);
    if (!GLOBAL) {
        if (threadIdx.x == 0) count_l = 0;
        __syncthreads();
    }
    double a[NR][NV], mu[NV];
    unsigned deadmask = 0;                          // bit k: column of slot k is eliminated (or padding)
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int col = BASQ_PAIRCOL(k);
        mu[k] = (col < M) ? mu_g[col] : 0.0;
        if (col >= M) deadmask |= 1u << k;
    }
#pragma unroll
    for (int jr = 0; jr < NR; ++jr) {
        const int c = gw + W * jr;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int col = BASQ_PAIRCOL(k);
            a[jr][k] = (c < nrows && col < M) ? PhiT_g[(size_t)c * M + col] : 0.0;
        }
    }
    // ratio test of one null vector (:148-152) + publication as pivot `kp`
    auto test_and_publish = [&](const double (&r)[NV], int kp) {
        double best = INF, brphi = 0.0, bphi = 0.0;
        int bcol = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const bool pos = !((deadmask >> k) & 1u) && (r[k] > 0.0);
            const double av = pos ? (mu[k] / r[k]) : INF;
            const double rp = pos ? (1.0 / r[k]) : 0.0;           // RN(1/phi): the pivot's reciprocal, one IEEE divide
            if (pos && (av < best || bcol == 0x7fffffff)) { best = av; bcol = BASQ_PAIRCOL(k); brphi = rp; bphi = r[k]; }
        }
        const double wmin = wave_min_f64(best);
        const int j = wave_min_i32((bcol != 0x7fffffff && best == wmin) ? bcol : 0x7fffffff);   // first index (:152)
        const int lane_j = (j & 127) >> 1;
        const double rphij = (j == 0x7fffffff) ? 0.0 : readlane_f64(brphi, lane_j);
        const double phij = (j == 0x7fffffff) ? 0.0 : readlane_f64(bphi, lane_j);
        const double hv = (lane == 0) ? wmin : (lane == 1) ? rphij : (lane == 2) ? phij : __longlong_as_double((long long)j);
        if (GLOBAL) {
            const unsigned gbase = (unsigned)(kp % D) * SLOT, tag = (unsigned)(kp + 1);
#pragma unroll
            for (int k = 0; k < NV; ++k) granule_store(grs, gbase + BASQ_PAIRCOL(k), tag, r[k], local);
            if (lane < 4) granule_store(grs, gbase + NC + lane, tag, hv, local);
        } else {
            double* slot = ring + (size_t)(kp % D) * SLOT;
#pragma unroll
            for (int k = 0; k < NV; ++k) slot[BASQ_PAIRCOL(k)] = r[k];
            if (lane < 4) slot[NC + lane] = hv;
            counter_publish(count, kp + 1, lane);
        }
    };
    if (nrows > 0 && gw == 0) test_and_publish(a[0], 0);
    int status = 0;
    for (int k = 0; k < nrows; ++k) {
        double phi[NV], hdr[4];
        if (GLOBAL) {
            const unsigned gbase = (unsigned)(k % D) * SLOT, tag = (unsigned)(k + 1);
            basq_v4u g[NV], gh;
            unsigned spins = 0;
            bool bad = false;
            // Only the owner of row k+1 is on the critical path: it sweeps the whole slot at once.  Thirty-one waves doing
            // the same (9 KB per sweep each) slow the very store they wait for (1.05 vs 0.59 ms at 200 x 400); they watch the
            // four header granules -- one 64-byte request per sweep -- and fetch the row once those carry the tag.
            if (!(k + 1 < nrows && gw == (k + 1) % W)) {
                for (;;) {
                    gh = granule_load(grs, gbase + NC + (lane & 3));
                    if (__all(granule_ok(gh, tag))) break;
                    if (++spins > BASQ_GRANULE_SPIN_LIMIT) { bad = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            for (; !bad;) {
#pragma unroll
                for (int kk = 0; kk < NV; ++kk) g[kk] = granule_load(grs, gbase + BASQ_PAIRCOL(kk));
                gh = granule_load(grs, gbase + NC + (lane & 3));
                bool ok = granule_ok(gh, tag);
#pragma unroll
                for (int kk = 0; kk < NV; ++kk) ok = ok && granule_ok(g[kk], tag);
                if (__all(ok)) break;
                if (++spins > BASQ_GRANULE_SPIN_LIMIT) { bad = true; break; }   // never in a healthy run (wave-uniform)
                if (spins > 16) __builtin_amdgcn_s_sleep(2);
            }
            if (bad) { status = 2; break; }       // this wave publishes nothing more: its siblings run into the same limit
#pragma unroll
            for (int kk = 0; kk < NV; ++kk) phi[kk] = granule_value(g[kk]);
            const double hv = granule_value(gh);
#pragma unroll
            for (int u = 0; u < 4; ++u) hdr[u] = readlane_f64(hv, u);
        } else {
            const int seen = counter_wait_gt(count, k);
            if (seen >= BASQ_ABORT_COUNT) { status = 2; break; }
            const double* slot = ring + (size_t)(k % D) * SLOT;
#pragma unroll
            for (int kk = 0; kk < NV; ++kk) phi[kk] = slot[BASQ_PAIRCOL(kk)];
#pragma unroll
            for (int u = 0; u < 4; ++u) hdr[u] = slot[NC + u];
        }
        const double aj = hdr[0], rphij = hdr[1], phij = hdr[2];
        const int j = __builtin_amdgcn_readfirstlane((int)__double_as_longlong(hdr[3]));
        if (j == 0x7fffffff) { status = 1; break; }               // uniform: no positive entry (the reference raises)
        const int kj = (j & 1) + 2 * (j >> 7), lane_j = (j & 127) >> 1;
#pragma unroll
        for (int kk = 0; kk < NV; ++kk) {                                               // :158-159
            const double step = aj * phi[kk];
            if (kk == kj && lane == lane_j) deadmask |= 1u << kk;
            mu[kk] = ((deadmask >> kk) & 1u) ? 0.0 : (mu[kk] - step);   // eliminated columns: the reference has Phi = 0, mu = 0
        }
        auto update_row = [&](double (&r)[NV]) {                                        // :165-171
            const double pj = readlane_f64(pick_slot<NV>(r, kj), lane_j);
#pragma unroll
            for (int kk = 0; kk < NV; ++kk) {
                const double o = div_by_recip(pj * phi[kk], phij, rphij);               // == (pj * phi) / phij, bit for bit
                r[kk] = r[kk] - o;
            }
        };
        if (k + 1 < nrows && gw == (k + 1) % W) {                  // my row is next: update it first, test, publish
            double rt[NV];
#pragma unroll
            for (int jr = 0; jr < NR; ++jr)
                if (gw + W * jr == k + 1) {
                    update_row(a[jr]);
#pragma unroll
                    for (int kk = 0; kk < NV; ++kk) rt[kk] = a[jr][kk];
                }
            test_and_publish(rt, k + 1);
        }
#pragma unroll
        for (int jr = 0; jr < NR; ++jr) {
            const int c = gw + W * jr;
            if (c > k + 1 && c < nrows) update_row(a[jr]);         // wave-uniform
        }
    }
    if (gw == 0) {
        // survivors: mu > 0 (:173-174), ascending column order; slot pair (2h, 2h+1) covers columns [128 h, 128 h + 128)
        unsigned long long bal[NV];
        bool keep[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            keep[k] = (BASQ_PAIRCOL(k) < M) && (mu[k] > 0.0);
            bal[k] = __ballot(keep[k]);
        }
        const unsigned long long below = (1ull << lane) - 1ull;
        int base = 0, total = 0;
#pragma unroll
        for (int k = 0; k < NV; ++k) total += __popcll(bal[k]);
#pragma unroll
        for (int h = 0; h < NV / 2; ++h) {
            const int r0 = base + __popcll(bal[2 * h] & below) + __popcll(bal[2 * h + 1] & below);
            const int r1 = r0 + (keep[2 * h] ? 1 : 0);
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int k = 2 * h + b, col = BASQ_PAIRCOL(k), rank = b ? r1 : r0;
                if (col < M) {
                    keep_rank[col] = keep[k] ? rank : -1;
                    if (keep[k]) { kept[rank] = col; w_star[rank] = mu[k]; }   // (mu_g stays as it came: ABI 13)
                }
            }
            base += __popcll(bal[2 * h]) + __popcll(bal[2 * h + 1]);
        }
        if (lane == 0) { info[0] = total; info[1] = status; }
    }
}

// ------------------------------------------------------------------------------------------------
// Caratheodory elimination (BASQ/_rchq.py:146-175), one work-group, null vectors in REGISTERS, handed over in BLOCKS
// (round 4; M <= 256, M - s <= 16 NR, (M - s)(M + 4) doubles of LDS).  Same arithmetic, op for op, as the kernels above.
//   * wave w owns the CONSECUTIVE null vectors NR w .. NR w + NR - 1 (lane l holds columns l, l + 64, l + 128, l + 192) and
//     its own copy of the weights.  It first CONSUMES the pivots of the earlier blocks -- one rank-1 update of its NR rows
//     per pivot, the pivot column's entry by v_readlane, nothing but the published row read from LDS -- and then PRODUCES
//     its block: ratio test on its next row, publish {row, j, alpha, 1 / phi_j, phi_j}, update its remaining rows -- a
//     dependent chain that stays inside one wave for NR steps; a wave whose block is done leaves;
//   * every pivot row is published ONCE into a slot of its own (the ring is the whole sequence: no slot is ever
//     reused, so there is no flow control) behind one monotone counter; 16 B x M per step of LDS traffic instead of the
//     24 B x M x (live rows) of car_eliminate_lds_kernel, which that kernel is bound by;
//   * the producer and the wave that produces next run at raised priority (s_setprio): the consumers' updates fill the
//     fp64 pipe, the chain of ratio tests must not queue behind them.
// ------------------------------------------------------------------------------------------------
#ifndef BASQ_RING_NR
#define BASQ_RING_NR 7           // rows per wave x waves of car_eliminate_ring_kernel's larger form (A/B builds: 9 x 12, 13 x 8, ...)
#endif
#ifndef BASQ_RING_WPG
#define BASQ_RING_WPG 16
#endif
template <bool TIGHT>
__device__ __forceinline__ int ring_wait_gt(int* cnt, int k) {
    unsigned spins = 0;
    int c;
    for (;;) {
        c = __hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
        c = __builtin_amdgcn_readfirstlane(c);
        if (c > k) break;
        if (++spins > BASQ_SPIN_LIMIT) {                           // never in a healthy run: every wave of the group gives up
            __hip_atomic_store(cnt, BASQ_ABORT_COUNT, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            c = BASQ_ABORT_COUNT;
            break;
        }
        if (!TIGHT) __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    return c;
}

// RN(1 / b) and RN(a / b) by the instruction sequence of the IEEE expansion WITHOUT its scaling steps (v_div_scale / v_div_fmas /
// v_div_fixup): the same bits whenever no scaling is due, i.e. for operands and quotients far from the ends of the exponent
// range -- the domain div_by_recip already assumes.  No VCC hand-over, so independent divisions interleave.
__device__ __forceinline__ double rcp_newton(double b) {
    double y = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-b, y, 1.0);
    return __builtin_fma(y, e, y);
}
__device__ __forceinline__ double div_newton(double a, double b, double y) {   // y = rcp_newton(b)
    const double q0 = a * y;
    const double r = __builtin_fma(-b, q0, a);
    return __builtin_fma(r, y, q0);
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
    const int BIG = -1;
    int x = (int)v;
    x = (int)min((unsigned)x, (unsigned)__builtin_amdgcn_update_dpp(BIG, x, 0x111, 0xf, 0xf, false));
    x = (int)min((unsigned)x, (unsigned)__builtin_amdgcn_update_dpp(BIG, x, 0x112, 0xf, 0xf, false));
    x = (int)min((unsigned)x, (unsigned)__builtin_amdgcn_update_dpp(BIG, x, 0x114, 0xf, 0xf, false));
    x = (int)min((unsigned)x, (unsigned)__builtin_amdgcn_update_dpp(BIG, x, 0x118, 0xf, 0xf, false));
    x = (int)min((unsigned)x, (unsigned)__builtin_amdgcn_update_dpp(BIG, x, 0x142, 0xa, 0xf, false));
    x = (int)min((unsigned)x, (unsigned)__builtin_amdgcn_update_dpp(BIG, x, 0x143, 0xc, 0xf, false));
    return (unsigned)__builtin_amdgcn_readlane(x, 63);
}
// minimum of a wave's doubles (no NaNs) through their order-preserving 64-bit keys: two 32-bit DPP reductions (each a single
// v_min_u32 per stage) instead of six stages of 64-bit moves + v_min_f64
__device__ __forceinline__ double wave_min_key_f64(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned long long key = b ^ (((long long)b >> 63) | 0x8000000000000000ull);
    const unsigned hi = (unsigned)(key >> 32), lo = (unsigned)key;
    const unsigned hmin = wave_min_u32(hi);
    const unsigned lmin = wave_min_u32(hi == hmin ? lo : 0xffffffffu);
    const unsigned long long kmin = ((unsigned long long)hmin << 32) | lmin;
    const unsigned long long bmin = (kmin >> 63) ? (kmin ^ 0x8000000000000000ull) : ~kmin;
    return __longlong_as_double((long long)bmin);
}

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>()), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>());
}

template <int NR, int WPG>
__global__ void __launch_bounds__(WPG * 64) car_eliminate_ring_kernel(const double* __restrict__ PhiT_g,
                                                                  const double* __restrict__ mu_g, int M, int s,
                                                                  int* __restrict__ keep_rank, int* __restrict__ kept,
                                                                  double* __restrict__ w_star, int* __restrict__ info) {
#pragma clang fp contract(off)   // plain operators: the reference rounds after every mul / sub / div
    // A wave issues at most one instruction per 4 cycles, whatever its kind, and dependent fp64 operations wait for each other:
    // the chain of ratio tests is bound by the instructions (and their latencies) between two publications.  Hence: no
    // dead-column mask (an eliminated column's weight becomes NaN, which the minimum skips and `> 0` rejects); divisions
    // without the scaling steps (they interleave); the wave minimum on 32-bit keys; the argmin's index from four ballots on the
    // scalar unit; ONE wave-uniform branch per pivot (the v_readlane's of the pivot column); a row is published before the
    // reciprocal of its pivot exists (every consumer computes its own); no guards inside the block, so that the updates of the
    // producer's later rows fill the latency of its next test.
    constexpr int NV = 4;
    extern __shared__ __attribute__((aligned(16))) double ring[];   // [nrows][M + 4]: published rows + {alpha, phi_j, j, -}
    __shared__ int count_l;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nrows = M - s, stride = M + 4;
    const int row0 = wv * NR;
    const int w_last = (nrows - 1) / NR;                            // owner of the last null vector: writes the outcome
    const double INF = __builtin_huge_val(), DEAD = __builtin_nan("");
    if (threadIdx.x == 0) count_l = 0;
    double a[NR][NV], mu[NV];
    bool valid[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        valid[k] = lane + 64 * k < M;
        mu[k] = valid[k] ? mu_g[lane + 64 * k] : DEAD;              // padding: never chosen, never kept
    }
#pragma unroll
    for (int jr = 0; jr < NR; ++jr)
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int r = row0 + jr;
            a[jr][k] = (r < nrows && valid[k]) ? PhiT_g[(size_t)r * M + lane + 64 * k] : 0.0;   // (rows past the end: zeros, updated
        }                                                                                       //  like the others, never tested)
    __syncthreads();
    if (row0 >= nrows) return;                                      // no null vector of its own (no barrier below)
    int* count = &count_l;
    int status = 0;
    const int my_rows = (nrows - row0 < NR) ? (nrows - row0) : NR;  // >= 1, wave-uniform
    // the entries of rows FIRST.. in the pivot column lane_j + 64 kj: ONE wave-uniform branch per pivot
    auto pivot_column = [&](auto FIRSTc, int kj, int lane_j, double (&pj)[NR]) {
        constexpr int FIRST = decltype(FIRSTc)::value;
        switch (kj) {
            case 0:
#pragma unroll
                for (int jr = FIRST; jr < NR; ++jr) pj[jr] = readlane_f64(a[jr][0], lane_j);
                break;
            case 1:
#pragma unroll
                for (int jr = FIRST; jr < NR; ++jr) pj[jr] = readlane_f64(a[jr][1], lane_j);
                break;
            case 2:
#pragma unroll
                for (int jr = FIRST; jr < NR; ++jr) pj[jr] = readlane_f64(a[jr][2], lane_j);
                break;
            default:
#pragma unroll
                for (int jr = FIRST; jr < NR; ++jr) pj[jr] = readlane_f64(a[jr][3], lane_j);
                break;
        }
    };
    // one pivot applied to the weights and to rows FIRST.. of this wave (:158-171)
    auto apply = [&](auto FIRSTc, const double (&phi)[NV], const double (&pj)[NR], double aj, int kj, int lane_j, double phij) {
        constexpr int FIRST = decltype(FIRSTc)::value;
        const double rphij = div_newton(1.0, phij, rcp_newton(phij));   // RN(1/phi_j): the pivot's reciprocal
        const bool mine = lane == lane_j;
#pragma unroll
        for (int kk = 0; kk < NV; ++kk) {
            const double step = aj * phi[kk];
            mu[kk] = (mine && kk == kj) ? DEAD : (mu[kk] - step);   // (the reference: mu[j] = 0, Phi[j, :] = 0 -- never positive again)
        }
        // stage by stage over two rows x four columns, the stages fenced for the scheduler: eight independent operations between
        // two dependent ones (a wave issues one instruction per four cycles; an fp64 result takes longer than that to come back)
#pragma unroll
        for (int j0 = FIRST; j0 < NR; j0 += 2) {
            constexpr int G = 2 * NV;
            double t[G], q0[G], rr[G];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int jr = j0 + g / NV;
                if (jr < NR) t[g] = pj[jr] * phi[g % NV];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < G; ++g)
                if (j0 + g / NV < NR) q0[g] = t[g] * rphij;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < G; ++g)
                if (j0 + g / NV < NR) rr[g] = __builtin_fma(-phij, q0[g], t[g]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < G; ++g)
                if (j0 + g / NV < NR) q0[g] = __builtin_fma(rr[g], rphij, q0[g]);       // == (pj * phi) / phij, bit for bit (div_by_recip)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int jr = j0 + g / NV;
                if (jr < NR) a[jr][g % NV] = a[jr][g % NV] - q0[g];
            }
        }
    };
    using I0 = std::integral_constant<int, 0>;
    // ---- consume the pivots of the earlier blocks ----
    // (raising the later half of the waves -- the longer backlogs -- above the earlier half was tried: 123.2 vs 122.1 us)
    for (int k = 0; k < row0; ++k) {
        const int owner = k / NR;
        const bool next = owner + 1 == wv;                         // this wave produces next: it must not fall behind
        if (next) __builtin_amdgcn_s_setprio(2);
        // (a consumer on the producer's SIMD sleeping until that block is complete, to leave the SIMD to the chain of ratio tests:
        //  129.8 us against 122 -- the consumers' throughput is needed throughout)
        const int seen = next ? ring_wait_gt<true>(count, k) : ring_wait_gt<false>(count, k);
        if (seen >= BASQ_ABORT_COUNT) { status = 2; break; }
        BASQ_NS_STAMP(k, 5);
        const double* slot = ring + (size_t)k * stride;
        double phi[NV], hdr[3];
#pragma unroll
        for (int kk = 0; kk < NV; ++kk) phi[kk] = valid[kk] ? slot[lane + 64 * kk] : 0.0;
#pragma unroll
        for (int u = 0; u < 3; ++u) hdr[u] = slot[M + u];
        const double aj = hdr[0], phij = hdr[1];
        const int j = __builtin_amdgcn_readfirstlane((int)__double_as_longlong(hdr[2]));
        if (j == 0x7fffffff) { status = 1; break; }               // uniform: no positive entry (the reference raises)
        double pj[NR];
        pivot_column(I0(), j >> 6, j & 63, pj);
        apply(I0(), phi, pj, aj, j >> 6, j & 63, phij);
        BASQ_NS_STAMP(k, 6);
    }
    // ---- produce this wave's block ----
    if (status == 0) {
        __builtin_amdgcn_s_setprio(3);
        auto produce = [&](auto JRc) -> bool {                     // -> false: no positive entry (status 1)
            constexpr int JR = decltype(JRc)::value;
            const int kp = row0 + JR;
            BASQ_NS_STAMP(kp, 0);
            // ratio test (:148-152) on a[JR], which has every earlier pivot applied
            double av[NV];
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const double q = div_newton(mu[k], a[JR][k], rcp_newton(a[JR][k]));    // NaN for an eliminated / padding column
                av[k] = (a[JR][k] > 0.0) ? q : INF;
            }
            double best = fmin(fmin(av[0], av[1]), fmin(av[2], av[3]));
            best = (best == best) ? best : INF;                     // four eliminated columns: keep NaNs (of either sign) out of the keys
            const double aj = wave_min_key_f64(best);
            BASQ_NS_STAMP(kp, 1);
            const bool found = aj < INF;
            double* slot = ring + (size_t)kp * stride;
            if (!found) {
                if (lane < 3) slot[M + lane] = __longlong_as_double(0x7fffffffLL);
                counter_publish(count, kp + 1, lane);
                return false;
            }
            // first index of the minimum (torch.argmin, :152): column = lane + 64 k, so the lowest slot with a hit wins
            const unsigned long long b0 = __ballot(av[0] == aj), b1 = __ballot(av[1] == aj), b2 = __ballot(av[2] == aj),
                                     b3 = __ballot(av[3] == aj);
            const int kj = b0 ? 0 : b1 ? 1 : b2 ? 2 : 3;
            const unsigned long long bj = b0 ? b0 : b1 ? b1 : b2 ? b2 : b3;
            const int lane_j = (int)__builtin_ctzll(bj);
            const int j = 64 * kj + lane_j;
            double pj[NR];
            pivot_column(JRc, kj, lane_j, pj);                     // pj[JR] = the pivot itself
            const double phij = pj[JR];
            BASQ_NS_STAMP(kp, 2);
#pragma unroll
            for (int k = 0; k < NV; ++k)
                if (valid[k]) slot[lane + 64 * k] = a[JR][k];
            const double hv = (lane == 0) ? aj : (lane == 1) ? phij : __longlong_as_double((long long)j);
            if (lane < 3) slot[M + lane] = hv;
            counter_publish(count, kp + 1, lane);
            BASQ_NS_STAMP(kp, 3);
            double phic[NV];
#pragma unroll
            for (int k = 0; k < NV; ++k) phic[k] = a[JR][k];
            apply(std::integral_constant<int, JR + 1>(), phic, pj, aj, kj, lane_j, phij);
            BASQ_NS_STAMP(kp, 4);
            return true;
        };
        bool ok = true;
        static_for<NR>([&](auto JRc) {
            if (ok && my_rows > decltype(JRc)::value) ok = produce(JRc);
        });
        if (!ok) status = 1;
        __builtin_amdgcn_s_setprio(0);
    }
    if (wv != w_last) return;
    // survivors: mu > 0 (:173-174), ascending column order (column = lane + 64 k)
    unsigned long long bal[NV];
    bool keep[NV];
    int total = 0;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        keep[k] = mu[k] > 0.0;                                      // (false for NaN: eliminated and padding columns)
        bal[k] = __ballot(keep[k]);
        total += __popcll(bal[k]);
    }
    const unsigned long long below = (1ull << lane) - 1ull;
    int base = 0;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int col = lane + 64 * k, rank = base + __popcll(bal[k] & below);
        if (col < M) {
            keep_rank[col] = keep[k] ? rank : -1;
            if (keep[k]) { kept[rank] = col; w_star[rank] = mu[k]; }
        }
        base += __popcll(bal[k]);
    }
    if (lane == 0) { info[0] = total; info[1] = status; }
}

// The same elimination for null vectors that one CU cannot hold (256 < M <= 448, M - s <= 256; n = 200: 200 x 400): several
// work-groups of 8 waves (two per SIMD, 256 registers each), wave g of the grid owns the consecutive rows 4 g .. 4 g + 3, and
// the pivot rows travel as tagged 16-byte granules in GLOBAL memory, one slot per pivot (no reuse: a launch zeroes the words
// once and a tag is its pivot's number + 1).  The chain of ratio tests stays inside a wave for four steps and crosses to the
// next wave through L2 -- where car_eliminate_cluster_kernel pays a trip through L2 on EVERY step; a consumer only ever waits
// for EARLIER blocks, so the work-groups need not be co-resident.  Same arithmetic, op for op.
#ifndef BASQ_GRING_NR
#define BASQ_GRING_NR 4          // rows per wave and waves per work-group of car_eliminate_gring_kernel (A/B builds)
#endif
#ifndef BASQ_GRING_WPG
#define BASQ_GRING_WPG 8
#endif
template <int NV, int NR, int WPG>
__global__ void __launch_bounds__(WPG * 64) car_eliminate_gring_kernel(const double* __restrict__ PhiT_g,
                                                                  const double* __restrict__ mu_g, int M, int s,
                                                                  int* __restrict__ keep_rank, int* __restrict__ kept,
                                                                  double* __restrict__ w_star, int* __restrict__ info, double* ws,
                                                                  int n_groups, int cluster_stride) {
#pragma clang fp contract(off)   // plain operators: the reference rounds after every mul / sub / div
    constexpr int NC = NV * 64, SLOT = NC + 4;
    if (blockIdx.x % cluster_stride) return;          // members share `blockIdx.x % 8`: one XCD under round-robin placement
    const int cu = blockIdx.x / cluster_stride;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int gw = cu * WPG + wv;
    const int nrows = M - s;
    const int row0 = gw * NR;
    const int w_last = (nrows - 1) / NR;
    const double INF = __builtin_huge_val(), DEAD = __builtin_nan("");
    // placement check (speed only): members on one XCD hand granules over with plain stores, which stay in that XCD's L2
    __shared__ int local_l;
    // (an LDS mailbox for the hand-over inside a work-group -- the next producer reading its predecessor's pivots from LDS
    //  instead of L2 -- was built twice and measured: 334 us against 307 at 200 x 400 with a release store of its counter, 341
    //  with a relaxed one; the hand-over is not what the step waits for.  The same kernel at 100 x 200, where one work-group
    //  can hold the null vectors: 118.8 us against car_eliminate_ring_kernel's 122.3 -- not worth a workspace and a memset there)
    if (threadIdx.x < 64) {
        const unsigned mine = 0x100u | (__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xfu);   // HW_REG_XCC_ID[3:0]
        unsigned* words = (unsigned*)ws;
        if (lane == 0) __hip_atomic_store((basq_gu32*)(words + cu), mine, BASQ_RLX_AGENT);
        unsigned f = mine, spins = 0;
        for (;;) {
            f = (lane < n_groups) ? __hip_atomic_load((basq_gu32*)(words + lane), BASQ_RLX_AGENT) : mine;
            if (__all(f != 0u) || ++spins > (1u << 12)) break;    // (a member that is not running yet counts as elsewhere)
            __builtin_amdgcn_s_sleep(1);
        }
        const bool same = __all(f == mine);                       // voted by all 64 lanes, OUTSIDE the lane-0 branch
        if (lane == 0) local_l = same ? 1 : 0;
    }
    __syncthreads();
    const bool local = local_l != 0;
    __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc((void*)(ws + 16), 0, (int)((size_t)nrows * SLOT * 16), 0x00020000);
    double a[NR][NV], mu[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) mu[k] = (lane + 64 * k < M) ? mu_g[lane + 64 * k] : DEAD;     // padding: never chosen, never kept
#pragma unroll
    for (int jr = 0; jr < NR; ++jr)
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int r = row0 + jr, col = lane + 64 * k;
            a[jr][k] = (r < nrows && col < M) ? PhiT_g[(size_t)r * M + col] : 0.0;
        }
    if (row0 >= nrows) return;
    int status = 0;
    const int my_rows = (nrows - row0 < NR) ? (nrows - row0) : NR;
    auto pivot_column = [&](auto FIRSTc, int kj, int lane_j, double (&pj)[NR]) {
        constexpr int FIRST = decltype(FIRSTc)::value;
#pragma unroll
        for (int kc = 0; kc < NV; ++kc)
            if (kj == kc) {                                        // wave-uniform
#pragma unroll
                for (int jr = FIRST; jr < NR; ++jr) pj[jr] = readlane_f64(a[jr][kc], lane_j);
            }
    };
    auto apply = [&](auto FIRSTc, const double (&phi)[NV], const double (&pj)[NR], double aj, int kj, int lane_j, double phij) {
        constexpr int FIRST = decltype(FIRSTc)::value;
        const double rphij = div_newton(1.0, phij, rcp_newton(phij));   // RN(1/phi_j): the pivot's reciprocal
        const bool mine = lane == lane_j;
#pragma unroll
        for (int kk = 0; kk < NV; ++kk) {
            const double step = aj * phi[kk];
            mu[kk] = (mine && kk == kj) ? DEAD : (mu[kk] - step);
        }
#pragma unroll
        for (int jr = FIRST; jr < NR; ++jr)
#pragma unroll
            for (int kk = 0; kk < NV; ++kk) {
                const double o = div_by_recip(pj[jr] * phi[kk], phij, rphij);           // == (pj * phi) / phij, bit for bit
                a[jr][kk] = a[jr][kk] - o;
            }
    };
    using I0 = std::integral_constant<int, 0>;
    // ---- consume the pivots of the earlier blocks ----
    for (int k = 0; k < row0; ++k) {
        const bool next = (k / NR) + 1 == gw;                      // this wave produces next
        if (next) __builtin_amdgcn_s_setprio(2);
        double phi[NV], aj, phij;
        int j;
        const unsigned gbase = (unsigned)k * SLOT, tag = (unsigned)(k + 1);
        basq_v4u g[NV], gh;
        unsigned spins = 0;
        bool bad = false, gave_up = false;
        if (!next) {                                               // far from its turn: watch the header only (one 64-byte request)
            for (;;) {
                gh = granule_load(grs, gbase + NC + (lane & 3));
                if (__all(granule_ok(gh, tag))) break;
                if (++spins > BASQ_GRANULE_SPIN_LIMIT) { bad = true; break; }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        for (; !bad;) {
#pragma unroll
            for (int kk = 0; kk < NV; ++kk) g[kk] = granule_load(grs, gbase + lane + 64 * kk);
            gh = granule_load(grs, gbase + NC + (lane & 3));
            bool ok = granule_ok(gh, tag);
            if (__all(ok)) {                                        // a producer that found no positive entry publishes the header only
                const int jh = __builtin_amdgcn_readfirstlane((int)__double_as_longlong(readlane_f64(granule_value(gh), 2)));
                if (jh == 0x7fffffff) { gave_up = true; break; }
            }
#pragma unroll
            for (int kk = 0; kk < NV; ++kk) ok = ok && granule_ok(g[kk], tag);
            if (__all(ok)) break;
            if (++spins > BASQ_GRANULE_SPIN_LIMIT) { bad = true; break; }   // never in a healthy run (wave-uniform)
            if (spins > 64) __builtin_amdgcn_s_sleep(1);
        }
        if (bad) { status = 2; break; }
        if (gave_up) { status = 1; break; }                       // uniform: no positive entry (the reference raises)
#pragma unroll
        for (int kk = 0; kk < NV; ++kk) phi[kk] = granule_value(g[kk]);
        const double hv = granule_value(gh);
        aj = readlane_f64(hv, 0);
        phij = readlane_f64(hv, 1);
        j = __builtin_amdgcn_readfirstlane((int)__double_as_longlong(readlane_f64(hv, 2)));
        if (j == 0x7fffffff) { status = 1; break; }               // uniform: no positive entry (the reference raises)
        double pj[NR];
        pivot_column(I0(), j >> 6, j & 63, pj);
        apply(I0(), phi, pj, aj, j >> 6, j & 63, phij);
    }
    // ---- produce this wave's block ----
    if (status == 0) {
        __builtin_amdgcn_s_setprio(3);
        auto produce = [&](auto JRc) -> bool {
            constexpr int JR = decltype(JRc)::value;
            const int kp = row0 + JR;
            const unsigned gbase = (unsigned)kp * SLOT, tag = (unsigned)(kp + 1);
            double av[NV], best = INF;
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const double q = div_newton(mu[k], a[JR][k], rcp_newton(a[JR][k]));    // NaN for an eliminated / padding column
                av[k] = (a[JR][k] > 0.0) ? q : INF;
                best = fmin(best, av[k]);
            }
            best = (best == best) ? best : INF;
            const double aj = wave_min_key_f64(best);
            if (!(aj < INF)) {
                if (lane < 4) granule_store(grs, gbase + NC + lane, tag, __longlong_as_double(0x7fffffffLL), local);
                return false;
            }
            int kj = -1;
            unsigned long long bj = 0;
#pragma unroll
            for (int k = 0; k < NV; ++k) {                         // first index of the minimum: the lowest slot with a hit
                const unsigned long long b = __ballot(av[k] == aj);
                if (kj < 0 && b) { kj = k; bj = b; }
            }
            const int lane_j = (int)__builtin_ctzll(bj);
            const int j = 64 * kj + lane_j;
            double pj[NR];
            pivot_column(JRc, kj, lane_j, pj);                     // pj[JR] = the pivot itself
            const double phij = pj[JR];
            const double hv = (lane == 0) ? aj : (lane == 1) ? phij : __longlong_as_double((long long)j);
#pragma unroll
            for (int k = 0; k < NV; ++k) granule_store(grs, gbase + lane + 64 * k, tag, a[JR][k], local);
            if (lane < 4) granule_store(grs, gbase + NC + lane, tag, hv, local);
            double phic[NV];
#pragma unroll
            for (int k = 0; k < NV; ++k) phic[k] = a[JR][k];
            apply(std::integral_constant<int, JR + 1>(), phic, pj, aj, kj, lane_j, phij);
            return true;
        };
        bool ok = true;
        if (ok && my_rows > 0) ok = produce(std::integral_constant<int, 0>());
        if constexpr (NR > 1) { if (ok && my_rows > 1) ok = produce(std::integral_constant<int, 1>()); }
        if constexpr (NR > 2) { if (ok && my_rows > 2) ok = produce(std::integral_constant<int, 2>()); }
        if constexpr (NR > 3) { if (ok && my_rows > 3) ok = produce(std::integral_constant<int, 3>()); }
        if constexpr (NR > 4) { if (ok && my_rows > 4) ok = produce(std::integral_constant<int, 4>()); }
        if constexpr (NR > 5) { if (ok && my_rows > 5) ok = produce(std::integral_constant<int, 5>()); }
        if constexpr (NR > 6) { if (ok && my_rows > 6) ok = produce(std::integral_constant<int, 6>()); }
        if constexpr (NR > 7) { if (ok && my_rows > 7) ok = produce(std::integral_constant<int, 7>()); }
        static_assert(NR <= 8, "unrolled by hand up to 8 rows per wave");
        if (!ok) status = 1;
        __builtin_amdgcn_s_setprio(0);
    }
    if (gw != w_last) return;
    // survivors: mu > 0 (:173-174), ascending column order (column = lane + 64 k)
    unsigned long long bal[NV];
    bool keep[NV];
    int total = 0;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        keep[k] = mu[k] > 0.0;
        bal[k] = __ballot(keep[k]);
        total += __popcll(bal[k]);
    }
    const unsigned long long below = (1ull << lane) - 1ull;
    int base = 0;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int col = lane + 64 * k, rank = base + __popcll(bal[k] & below);
        if (col < M) {
            keep_rank[col] = keep[k] ? rank : -1;
            if (keep[k]) { kept[rank] = col; w_star[rank] = mu[k]; }
        }
        base += __popcll(bal[k]);
    }
    if (lane == 0) { info[0] = total; info[1] = status; }
}

// ------------------------------------------------------------------------------------------------
// Null space of the wide [m, n] Caratheodory matrix (BASQ/_rchq.py:140-143) without an SVD iteration.
//
// The reference takes Phi = Vh[-(n-m):].T from torch.linalg.svd(X) (LAPACK gesdd).  gesdd first reduces X to
// lower-bidiagonal form B = Q^T X P with Householder reflectors (dgebrd, m < n), then diagonalises B by rotations
// that only mix the FIRST m rows of P^T; rows m..n-1 of Vh are therefore rows m..n-1 of
//     P^T = (G_0 G_1 ... G_{m-1})^T,   G_i = I - tau_i v_i v_i^T   (dlarfg convention, v_i = [0.., 1, x/(alpha-beta)])
// -- signs included.  The elimination's pivots depend on this very basis (not just on the null space), so the
// same reflectors are generated here, in LAPACK's order (dgebd2: right reflector from row i, apply to the rows
// below; left reflector from column i, apply to the trailing block), and tests pin the result to the host SVD.
//
// bidiag_reflectors_kernel: one work-group.  Rows have FIXED owners (row r -> wave r % NW), lanes own the columns
// c = lane + 64k, so a row never leaves its wave: the right reflector's A v and rank-1 update are wave-local
// (registers + shuffles), only the left reflector's u^T A needs a cross-wave sum (NW partial rows in LDS, summed
// in wave order).  The first NREG*NW rows live in registers, the rest in LDS (or in the V buffer in global memory
// when they do not fit); 4 barriers per step.  Outputs V[i, :] = v_i and tau[i].
// nullspace_apply_kernel: one wave per null vector c: y = e_{m+c}; for i = m-1..0: y -= tau_i (v_i . y) v_i.
// ------------------------------------------------------------------------------------------------
// dlarfg: reflector for (alpha, x) from alpha and |x|^2; returns tau, scale = 1/(alpha - beta) (0, 0 if x == 0)
__device__ __forceinline__ void householder_params(double alpha, double ss, double& tau, double& scale) {
    if (ss == 0.0) { tau = 0.0; scale = 0.0; return; }
    const double nrm = __builtin_sqrt(alpha * alpha + ss);
    const double beta = (alpha >= 0.0) ? -nrm : nrm;
    tau = (beta - alpha) / beta;
    scale = 1.0 / (alpha - beta);
}

template <int NV, int NREG, int NW, bool ROWS_IN_LDS>
__global__ void __launch_bounds__(NW * 64) bidiag_reflectors_kernel(const double* __restrict__ X, int m, int n,
                                                                    double* __restrict__ V, double* __restrict__ tau_g) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    __shared__ double tau_sh;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double* vsh = sm;                          // [n]  current right reflector
    double* wsh = vsh + n;                     // [n]  u^T A
    double* ush = wsh + n;                     // [m]  column i below the diagonal
    double* wpart = ush + m;                   // [NW][n] per-wave partials of u^T A
    double* lrows = wpart + (size_t)NW * n;    // rows r >= NREG*NW (ROWS_IN_LDS) at (r - NREG*NW) * n
    constexpr int RBASE = NREG * NW;
    double reg[NREG > 0 ? NREG : 1][NV];

#define BASQ_ROW_LD(r, c) (ROWS_IN_LDS ? lrows[(size_t)((r) - RBASE) * n + (c)] : V[(size_t)(r) * n + (c)])
#define BASQ_ROW_ST(r, c, val)                                                     \
    do {                                                                           \
        if (ROWS_IN_LDS) lrows[(size_t)((r) - RBASE) * n + (c)] = (val);           \
        else V[(size_t)(r) * n + (c)] = (val);                                     \
    } while (0)
// BODY sees (int r, double a[NV]); rows r >= r0 owned by this wave; memory rows are written back when WRITE
#define BASQ_OWN_ROWS(r0, WRITE, BODY)                                             \
    do {                                                                           \
        _Pragma("unroll") for (int jr = 0; jr < NREG; ++jr) {                      \
            const int r = wv + jr * NW;                                            \
            if (r >= (r0) && r < m) {                                              \
                double(&a)[NV] = reg[jr];                                          \
                BODY                                                               \
            }                                                                      \
        }                                                                          \
        for (int r = wv + RBASE; r < m; r += NW) {                                 \
            if (r < (r0)) continue;                                                \
            double a[NV];                                                          \
            _Pragma("unroll") for (int k = 0; k < NV; ++k) {                       \
                const int c = lane + 64 * k;                                       \
                a[k] = (c < n) ? BASQ_ROW_LD(r, c) : 0.0;                          \
            }                                                                      \
            BODY                                                                   \
            if (WRITE) {                                                           \
                _Pragma("unroll") for (int k = 0; k < NV; ++k) {                   \
                    const int c = lane + 64 * k;                                   \
                    if (c < n) BASQ_ROW_ST(r, c, a[k]);                            \
                }                                                                  \
            }                                                                      \
        }                                                                          \
    } while (0)

    // load: every wave fetches its own rows (no other wave ever touches them)
#pragma unroll
    for (int jr = 0; jr < NREG; ++jr) {
        const int r = wv + jr * NW;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int c = lane + 64 * k;
            reg[jr][k] = (r < m && c < n) ? X[(size_t)r * n + c] : 0.0;
        }
    }
    for (int r = wv + RBASE; r < m; r += NW)
        for (int c = lane; c < n; c += 64) BASQ_ROW_ST(r, c, X[(size_t)r * n + c]);

    for (int i = 0; i < m; ++i) {
        const int ik = i >> 6, il = i & 63;
        // ---- S1: right reflector G_i from row i (owner wave only) ----
        if (wv == i % NW) {
            double a[NV];
            if (i < RBASE) {
#pragma unroll
                for (int jr = 0; jr < NREG; ++jr)
                    if (i == wv + jr * NW) {
#pragma unroll
                        for (int k = 0; k < NV; ++k) a[k] = reg[jr][k];
                    }
            } else {
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    const int c = lane + 64 * k;
                    a[k] = (c < n) ? BASQ_ROW_LD(i, c) : 0.0;
                }
            }
            double ss = 0.0, al = 0.0;
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int c = lane + 64 * k;
                if (c > i && c < n) ss += a[k] * a[k];
                if (k == ik) al = a[k];
            }
            ss = wave_sum(ss);
            const double alpha = __shfl(al, il, 64);
            double tau, scale;
            householder_params(alpha, ss, tau, scale);
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int c = lane + 64 * k;
                if (c < n) {
                    const double v = (c < i) ? 0.0 : ((c == i) ? 1.0 : a[k] * scale);
                    vsh[c] = v;
                    V[(size_t)i * n + c] = v;
                }
            }
            if (lane == 0) { tau_sh = tau; tau_g[i] = tau; }
        }
        if (i == m - 1) break;
        __syncthreads();
        // ---- S2: A[i+1:, i:] -= tau (A v) v^T, wave-local per row; publish column i ----
        {
            const double tau = tau_sh;
            double vr[NV];
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int c = lane + 64 * k;
                vr[k] = (c < n) ? vsh[c] : 0.0;
            }
            BASQ_OWN_ROWS(i + 1, true, {
                double dot = 0.0;
                _Pragma("unroll") for (int k = 0; k < NV; ++k) dot += a[k] * vr[k];
                dot = wave_sum(dot);
                const double t = tau * dot;
                double ci = 0.0;
                _Pragma("unroll") for (int k = 0; k < NV; ++k) {
                    a[k] -= t * vr[k];
                    if (k == ik) ci = a[k];
                }
                if (lane == il) ush[r] = ci;
            });
        }
        __syncthreads();
        // ---- S3: left reflector H_i from column i (rows i+1..), partial u^T A per wave ----
        double tauq, scale2;
        {
            double ss = 0.0;
            for (int r = i + 2 + lane; r < m; r += 64) ss += ush[r] * ush[r];
            ss = wave_sum(ss);
            householder_params(ush[i + 1], ss, tauq, scale2);
            double pw[NV];
#pragma unroll
            for (int k = 0; k < NV; ++k) pw[k] = 0.0;
            BASQ_OWN_ROWS(i + 1, false, {
                const double ur = (r == i + 1) ? 1.0 : ush[r] * scale2;
                _Pragma("unroll") for (int k = 0; k < NV; ++k) pw[k] += ur * a[k];
            });
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int c = lane + 64 * k;
                if (c < n) wpart[(size_t)wv * n + c] = pw[k];
            }
        }
        __syncthreads();
        for (int c = tid; c < n; c += NW * 64) {
            double acc = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) acc += wpart[(size_t)w * n + c];
            wsh[c] = acc;
        }
        __syncthreads();
        // ---- S5: A[i+1:, i+1:] -= tauq u (u^T A) ----
        {
            double wc[NV];
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int c = lane + 64 * k;
                wc[k] = (c > i && c < n) ? wsh[c] : 0.0;
            }
            BASQ_OWN_ROWS(i + 1, true, {
                const double ur = (r == i + 1) ? 1.0 : ush[r] * scale2;
                const double t = tauq * ur;
                _Pragma("unroll") for (int k = 0; k < NV; ++k) a[k] -= t * wc[k];
            });
        }
        // no barrier: the next step's S1 touches only vsh / tau_sh / its own row; ush, wpart and wsh are rewritten
        // after the next barriers, when every wave has left S5.
    }
#undef BASQ_OWN_ROWS
#undef BASQ_ROW_ST
#undef BASQ_ROW_LD
}

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

// householder_params for the serial section of the kernel below, as one short dependency chain: with
// n = |(alpha, x)| and s = sign(alpha):  beta = -s n,  tau = (beta - alpha)/beta = 1 + |alpha| / n,
// scale = 1/(alpha - beta) = s / (|alpha| + n)   (n and 1/n from one Newton iteration, one reciprocal).
// Branch-free (a zero tail selects tau = scale = 0 at the end: the callers' serial chains carry no jump, and the loads behind the
// call are not held back by one).
__device__ __forceinline__ void householder_params_fast(double alpha, double ss, double& tau, double& scale) {
    double nrm, rnrm;
    sqrt_rsqrt_nr(__builtin_fma(alpha, alpha, ss), nrm, rnrm);
    const double aa = __builtin_fabs(alpha);
    const double t = __builtin_fma(aa, rnrm, 1.0);
    const double r = recip_nr(aa + nrm);
    const bool none = ss == 0.0;
    tau = none ? 0.0 : t;
    scale = none ? 0.0 : ((alpha >= 0.0) ? r : -r);
}

template <int CTRL>
__device__ __forceinline__ double dpp_perm_f64(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// Four 64-lane sums for the price of one (gfx950 v_permlane32_swap / v_permlane16_swap): fold the wave in half
// with x0,x1 (resp. x2,x3) sharing a register, fold the 16-lane rows with the two pairs sharing a register, then
// an xor-butterfly inside each row of 16.  Totals come back wave-uniform (SGPRs).
__device__ __forceinline__ void wave_sum4(double& x0, double& x1, double& x2, double& x3) {
    auto fold32 = [](double a, double b) {     // lanes 0-31: a[l] + a[l+32];  lanes 32-63: b[l-32] + b[l]
        const long long ba = __double_as_longlong(a), bb = __double_as_longlong(b);
        const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)ba, (unsigned)bb, false, false);
        const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(ba >> 32), (unsigned)(bb >> 32), false, false);
        const double a2 = __longlong_as_double(((long long)hi[0] << 32) | lo[0]);
        const double b2 = __longlong_as_double(((long long)hi[1] << 32) | lo[1]);
        return a2 + b2;
    };
    auto fold16 = [](double a, double b) {     // rows of 16: [a r0 + a r1, b r0 + b r1, a r2 + a r3, b r2 + b r3]
        const long long ba = __double_as_longlong(a), bb = __double_as_longlong(b);
        const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)ba, (unsigned)bb, false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(ba >> 32), (unsigned)(bb >> 32), false, false);
        const double a2 = __longlong_as_double(((long long)hi[0] << 32) | lo[0]);
        const double b2 = __longlong_as_double(((long long)hi[1] << 32) | lo[1]);
        return a2 + b2;
    };
    double v = fold16(fold32(x0, x1), fold32(x2, x3));   // rows: x0 | x2 | x1 | x3
    v += dpp_perm_f64<0xB1>(v);     // quad_perm [1,0,3,2]
    v += dpp_perm_f64<0x4E>(v);     // quad_perm [2,3,0,1]
    v += dpp_perm_f64<0x141>(v);    // row_half_mirror
    v += dpp_perm_f64<0x140>(v);    // row_mirror
    x0 = readlane_f64(v, 0);
    x2 = readlane_f64(v, 16);
    x1 = readlane_f64(v, 32);
    x3 = readlane_f64(v, 48);
}

// Register-resident form for m <= 16*NREG rows, n <= 64*NV columns (the headline 100 x 200 fits NREG = 7, NV = 4):
// the whole matrix lives in VGPRs (row r -> wave r % 16, slot r / 16; lanes own column PAIRS, so the broadcast
// vectors move as 16-byte LDS accesses), LDS carries only those vectors.  Three barriers per step t:
//   phase A (all waves, live rows r > t only): apply the PREVIOUS left reflector H_{t-1} (deferred), apply G_t
//       (row dots reduced four at a time, rank-1 update), read column t of the updated rows back through SGPRs,
//       accumulate this wave's share of column_t^T A and |column_t|^2; the owner of row t+1 publishes that row;
//   phase B, spread over NV waves (one per SIMD) with ONE COLUMN PER LANE (round 5; rounds 2-4 ran it on wave 0 alone, four
//       columns per lane: 2 400 of a step's 8 000 cycles with fifteen waves parked, profiles/r06_v_bidiag_phase_clock_100x200.txt):
//       B1: every B wave sums the 16 partial rows of ITS 64 columns in one LDS pass, works out H_t's tauq / u scale from the 16
//           partial norms (redundantly: no hand-over), forms w = u^T A and row t+1 after H_t on its columns, and reduces its share of
//           the new row's norm;
//       B2 (behind a barrier that carries the NV partial norms and alpha): G_{t+1}'s parameters (redundantly, bit-identical in
//           every B wave), v_{t+1} on its columns -> LDS and V.
// Same reflectors as dgebd2; only the association of the sums differs (agreement with LAPACK ~1e-14).
template <int NV, int NREG>
__global__ void __launch_bounds__(1024) bidiag_reflectors_reg_kernel(const double* __restrict__ X, int m, int n,
                                                                     double* __restrict__ V,
                                                                     double* __restrict__ tau_g) {
    static_assert(NV % 2 == 0, "lanes own column pairs");
    constexpr int NW = 16, NC = NV * 64, NG = (NREG + 3) / 4;
    __shared__ __attribute__((aligned(16))) double vsh[NC];          // v_t
    __shared__ __attribute__((aligned(16))) double wsh[NC];          // w of H_{t-1}, zero for c < t
    __shared__ __attribute__((aligned(16))) double r1sh[NC];         // row t+1 after G_t
    __shared__ __attribute__((aligned(16))) double wpart[NW * NC];   // per-wave partials of column_t^T A
    __shared__ __attribute__((aligned(16))) double sspart[NW];
    __shared__ double par[4];             // tau_t, tauq_{t-1}, u-scale_{t-1}, alpha of H_t
    __shared__ double ssb[NV + 1];        // phase B: the B waves' shares of |row t+1|^2 beyond its pivot, and the pivot itself
#ifdef BASQ_NS_PROF
    // phase clock of tools/ns_prof.hip: cycles per phase summed over all steps in SCALAR registers (the kernel sits at its VGPR
    // ceiling: stamps that touch a vector register make it spill ~370 of them and run 7 x slower)
    unsigned long long tacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev));
#define BASQ_NS_LSTAMP(t, slot)                                                                 \
    do {                                                                                       \
        unsigned long long now_;                                                               \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_));                      \
        tacc[slot] += now_ - tprev;                                                            \
        tprev = now_;                                                                          \
    } while (0)
#else
#define BASQ_NS_LSTAMP(t, slot) do { } while (0)
#endif
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave-uniform: a scalar)
    // slot k of this lane holds column COL(k); column c sits in lane (c & 127) >> 1, slot (c & 1) + 2 (c >> 7)
#define BASQ_COL(k) (2 * lane + ((k) & 1) + 128 * ((k) >> 1))
    double a[NG * 4][NV], cprev[NG * 4];
#pragma unroll
    for (int jr = 0; jr < NG * 4; ++jr) {
        const int r = wv + NW * jr;
        cprev[jr] = 0.0;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int c = BASQ_COL(k);
            a[jr][k] = (jr < NREG && r < m && c < n) ? X[(size_t)r * n + c] : 0.0;
        }
    }
    // G_t's second half, on the B waves (column cb = 64 wv + lane each): parameters from the NV partial norms + the pivot in
    // ssb[] (summed in index order: the same bits in every B wave), then v_t on the wave's columns -> vsh, V; tau -> par[0], tau_g
    auto publish_right = [&](double rn, int t) {
        const int cb = 64 * wv + lane;
        const double alpha = ssb[NV];
        double ss = ssb[0];
#pragma unroll
        for (int w = 1; w < NV; ++w) ss += ssb[w];
        double tau, scale;
        householder_params_fast(alpha, ss, tau, scale);
        const double v = (cb < t) ? 0.0 : ((cb == t) ? 1.0 : rn * scale);
        vsh[cb] = v;
        if (cb < n) V[(size_t)t * n + cb] = v;
        if (tid == 0) { par[0] = tau; tau_g[t] = tau; }
    };
    // ... and its first half: this wave's share of |row[t+1:]|^2 and the pivot row[t] -> ssb[]
    auto norm_share = [&](double rn, int t) {
        const int cb = 64 * wv + lane;
        const double ssl = wave_sum((cb > t) ? rn * rn : 0.0);
        if (lane == 0) ssb[wv] = ssl;
        if (cb == t) ssb[NV] = rn;
    };
    // prologue: G_0 from row 0 (wave 0's slot 0), through the same two halves
    if (wv == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            r1sh[BASQ_COL(k)] = a[0][k];
            wsh[BASQ_COL(k)] = 0.0;
        }
        if (lane == 0) { par[1] = 0.0; par[2] = 0.0; }
    }
    __syncthreads();
    double rn0 = 0.0;
    if (wv < NV) {
        rn0 = r1sh[64 * wv + lane];
        norm_share(rn0, 0);
    }
    __syncthreads();
    if (wv < NV) publish_right(rn0, 0);
    __syncthreads();
    for (int t = 0; t + 1 < m; ++t) {
        const int tk = (t & 1) + 2 * (t >> 7), tl = (t & 127) >> 1;
        BASQ_NS_LSTAMP(t, 0);
        {   // ---- phase A ----
            const double tau = par[0], kappa = par[1] * par[2];     // tauq * u-scale of H_{t-1}
            double vr[NV], wc[NV], pw[NV];
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                vr[k] = vsh[BASQ_COL(k)];
                wc[k] = wsh[BASQ_COL(k)];
                pw[k] = 0.0;
            }
            double ssp = 0.0;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (wv + NW * (4 * g + 3) <= t) continue;           // wave-uniform: the whole group is dead
                double dot[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int jr = 4 * g + j, r = wv + NW * jr;
                    dot[j] = 0.0;
                    if (jr < NREG && r > t && r < m) {              // wave-uniform
                        const double tu = kappa * cprev[jr];        // H_{t-1}: tauq u_r, u_r = column_{t-1}[r] * scale
#pragma unroll
                        for (int k = 0; k < NV; ++k) {
                            a[jr][k] -= tu * wc[k];
                            dot[j] += a[jr][k] * vr[k];
                        }
                    }
                }
                wave_sum4(dot[0], dot[1], dot[2], dot[3]);
                if (g == 0) BASQ_NS_LSTAMP(t, 1);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int jr = 4 * g + j, r = wv + NW * jr;
                    if (jr < NREG && r > t && r < m) {
                        const double td = tau * dot[j];             // G_t
#pragma unroll
                        for (int k = 0; k < NV; ++k) a[jr][k] -= td * vr[k];
                        // A[r][t] after G_t: slot tk (wave-uniform) of lane tl.  A scalar BRANCH per slot, not a select chain: eight
                        // v_cndmask per row were a fifth of this phase's vector instructions (the empty volatile asm keeps the
                        // compiler from converting the branches back into selects)
                        double cr = 0.0;
#pragma unroll
                        for (int k = 0; k < NV; ++k)
                            if (k == tk) {
                                cr = readlane_f64(a[jr][k], tl);
                                asm volatile("");
                            }
                        cprev[jr] = cr;
                        if (r == t + 1) {
#pragma unroll
                            for (int k = 0; k < NV; ++k) r1sh[BASQ_COL(k)] = a[jr][k];
                            if (lane == 0) par[3] = cr;
                        } else {
#pragma unroll
                            for (int k = 0; k < NV; ++k) pw[k] += cr * a[jr][k];
                            ssp += cr * cr;
                        }
                    }
                }
            }
            BASQ_NS_LSTAMP(t, 2);
#pragma unroll
            for (int k = 0; k < NV; ++k) wpart[wv * NC + BASQ_COL(k)] = pw[k];
            if (lane == 0) sspart[wv] = ssp;
        }
        BASQ_NS_LSTAMP(t, 3);
        __syncthreads();
        BASQ_NS_LSTAMP(t, 4);
        double rn = 0.0;
        if (wv < NV) {   // ---- phase B1: one column per lane ----
            const int cb = 64 * wv + lane;
            // the 16 partial rows of this column, wave order 0..15 (conflict-free: consecutive lanes, consecutive doubles); two
            // chains so that the adds do not wait for one another
            // |column_t|^2 from the 16 partial norms: lane l reads partial l & 15, prefix sums inside the rows of 16 lanes (row_shr
            // 1, 2, 4, 8), lane 15 read back -- one LDS trip + four DPP steps instead of sixteen dependent adds
            const double alphaH = par[3];
            double ss2 = sspart[lane & 15];
            ss2 += dpp_shift_f64<0x111, 0xf>(ss2);
            ss2 += dpp_shift_f64<0x112, 0xf>(ss2);
            ss2 += dpp_shift_f64<0x114, 0xf>(ss2);
            ss2 += dpp_shift_f64<0x118, 0xf>(ss2);
            ss2 = readlane_f64(ss2, 15);
            // the 16 partial rows of this column (read in groups of four behind compiler fences: left alone, the scheduler hoists all
            // reads above the sums and spills a dozen registers of the matrix, which lives in this wave's VGPRs throughout); they are
            // in flight while H_t's parameters go through their chain of dependent operations
            double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
            for (int w = 0; w < NW; w += 4) {
                acc0 += wpart[w * NC + cb];
                acc1 += wpart[(w + 1) * NC + cb];
                acc0 += wpart[(w + 2) * NC + cb];
                acc1 += wpart[(w + 3) * NC + cb];
                if (w + 4 < NW) asm volatile("" ::: "memory");
            }
            double tauq, uscale;
            householder_params_fast(alphaH, ss2, tauq, uscale);     // H_t (every B wave: the same bits, nothing handed over)
            BASQ_NS_LSTAMP(t, 5);
            const double r1 = r1sh[cb];
            const double w_c = (cb > t) ? __builtin_fma(uscale, acc0 + acc1, r1) : 0.0;   // u^T A with u = [1, column * scale]
            wsh[cb] = w_c;
            rn = r1 - tauq * w_c;                                                          // row t+1 after H_t
            norm_share(rn, t + 1);
            if (tid == 0) { par[1] = tauq; par[2] = uscale; }
            BASQ_NS_LSTAMP(t, 6);
        }
        __syncthreads();
        BASQ_NS_LSTAMP(t, 7);
        if (wv < NV) {   // ---- phase B2 ----
            publish_right(rn, t + 1);
            BASQ_NS_LSTAMP(t, 8);
        }
        __syncthreads();
    }
#ifdef BASQ_NS_PROF
    if (lane == 0)
        for (int i = 0; i < 10; ++i) g_ns_prof[wv * 10 + i] = (long long)tacc[i];
#endif
#undef BASQ_NS_LSTAMP
#undef BASQ_COL
}

// ------------------------------------------------------------------------------------------------
// Bidiagonalisation reflectors, cluster form (same reflectors as dgebd2 / bidiag_reflectors_reg_kernel; layout and
// exchange as described above car_eliminate_cluster_kernel).  ONE synchronisation per step t:
//   bulk (every wave, its live rows r > t): apply the previous left reflector H_{t-1} (deferred), the dot products
//       with v_t (reduced four rows at a time), the rank-1 update by G_t; column t of the updated rows is read back
//       and accumulated into this wave's share of column_t^T A and |column_t|^2; the owner of row t+1 publishes it;
//   exchange: the waves' partial rows through LDS + one s_barrier; clusters add one hop through global memory (the
//       work-group's sum, published write-through by wave 0 behind the barrier, one flag per work-group);
//   chain (EVERY wave, redundantly and bit-identically -- nothing is handed back): sum the partials in a fixed order,
//       H_t's parameters, w = u^T A, row t+1 after H_t, then G_{t+1} from it -> v_{t+1}, tau_{t+1}.
// ------------------------------------------------------------------------------------------------
template <int NV, int NR, int NCU>
__global__ void __launch_bounds__(BASQ_WPG * 64) bidiag_cluster_kernel(const double* __restrict__ X, int m, int n,
                                                             double* __restrict__ V, double* __restrict__ tau_g,
                                                             double* ws, int cluster_stride, int* __restrict__ info_g) {
    static_assert(NV % 2 == 0, "lanes own column pairs");
    constexpr int WPG = BASQ_WPG, W = WPG * NCU, NC = NV * 64, NG = (NR + 3) / 4, MSG = NC + 8;
    constexpr bool GLOBAL = NCU > 1;
    typedef double d2_t __attribute__((ext_vector_type(2)));
    if (blockIdx.x % cluster_stride) return;
    const int cu = blockIdx.x / cluster_stride;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int gw = cu * WPG + wv;
    __shared__ __attribute__((aligned(16))) double wpart_l[2 * WPG * MSG];   // [parity][local wave]: partial row | ssp
    __shared__ __attribute__((aligned(16))) double r1_l[2 * MSG];          // [parity]: row t+1 after G_t | its column-t entry
    __shared__ int abort_l;
    // global (clusters): ws = [16 unused doubles][2][NCU][MSG] granules of work-group sums [2][MSG] granules of row t+1
    static_assert(!GLOBAL || NC == WPG * 64, "the exchange gives every thread of a work-group one column");
    constexpr unsigned R1BASE = 2u * NCU * MSG;                      // granule index of the row t+1 messages
    __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc((void*)(ws + 16), 0,
                                                                   GLOBAL ? (int)((R1BASE + 2u * MSG) * 16u) : 0, 0x00020000);
    if (threadIdx.x == 0) abort_l = 0;
    __shared__ int local_l;
    bool local = false;
    if (GLOBAL) local = cluster_shares_xcd<NCU>((unsigned*)ws, cu, &local_l);

    double a[NG * 4][NV], cprev[NG * 4];
#pragma unroll
    for (int jr = 0; jr < NG * 4; ++jr) {
        const int r = gw + W * jr;
        cprev[jr] = 0.0;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int c = BASQ_PAIRCOL(k);
            a[jr][k] = (jr < NR && r < m && c < n) ? X[(size_t)r * n + c] : 0.0;
        }
    }
    double vr[NV], wc[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) { vr[k] = 0.0; wc[k] = 0.0; }
    double tau = 0.0, kappa = 0.0;
    bool aborted = false;
    for (int t = -1; t + 1 < m; ++t) {
        const int par = (t + 1) & 1;
        double* my_msg = wpart_l + (size_t)(par * WPG + wv) * MSG;
        double* r1buf = r1_l + (size_t)par * MSG;                   // (one work-group; clusters publish granules instead)
        const unsigned epoch = (unsigned)(t + 2), r1g = R1BASE + (unsigned)par * MSG;
        double pw[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) pw[k] = 0.0;
        double ssp = 0.0;
        BASQ_NS_STAMP(t + 1, 0);
        if (t >= 0) {   // ---- bulk ----
            const int tk = (t & 1) + 2 * (t >> 7), tl = (t & 127) >> 1;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (gw + W * (4 * g + 3) <= t) continue;           // wave-uniform: the whole group is dead
                double dot[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int jr = 4 * g + j, r = gw + W * jr;
                    dot[j] = 0.0;
                    if (jr < NR && r > t && r < m) {                // wave-uniform
                        const double tu = kappa * cprev[jr];        // H_{t-1}: tauq u_r, u_r = column_{t-1}[r] * scale
#pragma unroll
                        for (int k = 0; k < NV; ++k) {
                            a[jr][k] -= tu * wc[k];
                            dot[j] += a[jr][k] * vr[k];
                        }
                    }
                }
                wave_sum4(dot[0], dot[1], dot[2], dot[3]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int jr = 4 * g + j, r = gw + W * jr;
                    if (jr < NR && r > t && r < m) {
                        const double td = tau * dot[j];             // G_t
#pragma unroll
                        for (int k = 0; k < NV; ++k) a[jr][k] -= td * vr[k];
                        const double cr = readlane_f64(pick_slot<NV>(a[jr], tk), tl);   // A[r][t] after G_t
                        cprev[jr] = cr;
                        if (r == t + 1) {
#pragma unroll
                            for (int k = 0; k < NV; ++k) {
                                if (GLOBAL) granule_store(grs, r1g + BASQ_PAIRCOL(k), epoch, (BASQ_PAIRCOL(k) == NC - 1) ? cr : a[jr][k], local);
                                else r1buf[BASQ_PAIRCOL(k)] = a[jr][k];
                            }
                            if (!GLOBAL && lane == 0) r1buf[NC] = cr;
                        } else {
#pragma unroll
                            for (int k = 0; k < NV; ++k) pw[k] += cr * a[jr][k];
                            ssp += cr * cr;
                        }
                    }
                }
            }
        } else if (gw == 0) {   // prologue: row 0 as it stands is "row t+1"
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                if (GLOBAL) granule_store(grs, r1g + BASQ_PAIRCOL(k), epoch, a[0][k], local);
                else r1buf[BASQ_PAIRCOL(k)] = a[0][k];
            }
        }
#pragma unroll
        for (int h = 0; h < NV / 2; ++h)
            *reinterpret_cast<d2_t*>(my_msg + 2 * lane + 128 * h) = (d2_t){pw[2 * h], pw[2 * h + 1]};
        if (lane == 0) my_msg[NC] = ssp;
        BASQ_NS_STAMP(t + 1, 1);
        __syncthreads();
        BASQ_NS_STAMP(t + 1, 2);
        // ---- chain (every wave) ----
        double accs[NV], ss2 = 0.0, r1[NV], rn[NV], alphaH = 0.0;
#pragma unroll
        for (int k = 0; k < NV; ++k) accs[k] = 0.0;
        if (!GLOBAL) {
            if (t >= 0) {
#pragma unroll
                for (int w = 0; w < WPG; ++w) {                     // local partials, wave order
                    const double* src = wpart_l + (size_t)(par * WPG + w) * MSG;
#pragma unroll
                    for (int h = 0; h < NV / 2; ++h) {
                        const d2_t v = *reinterpret_cast<const d2_t*>(src + 2 * lane + 128 * h);
                        accs[2 * h] += v.x;
                        accs[2 * h + 1] += v.y;
                    }
                    ss2 += src[NC];
                }
            }
        }
        if (GLOBAL) {
            // Every thread of the work-group owns ONE column of the exchange: it adds the eight local partials (wave order),
            // publishes the work-group's sum as a tagged granule and sweeps the NCU granules of that column and the granule
            // of the published row t+1 until every tag carries this step's epoch -- one fabric trip, no drain, no flag.
            // The cluster-order sum and the row go to the sibling waves through LDS (one barrier).  The two scalars of a
            // step (|column|^2 partials, the pivot entry) ride in the last column slot, NC - 1, which the matrix never
            // uses (the launcher admits n <= NC - 2 only).
            const unsigned col = (unsigned)(wv * 64 + lane);
            double* tot_l = wpart_l + (size_t)(par * WPG) * MSG;    // this parity's first partial slot: a thread overwrites
            double s_own = 0.0;                                     // only the entry it has just read
            if (t >= 0) {
                const unsigned lcol = (col == NC - 1) ? NC : col;   // the scalar's slot in the local messages
#pragma unroll
                for (int w = 0; w < WPG; ++w) s_own += wpart_l[(size_t)(par * WPG + w) * MSG + lcol];   // wave order
                granule_store(grs, (unsigned)(par * NCU + cu) * MSG + col, epoch, s_own, local);
            }
            basq_v4u gq[NCU], g1;
            unsigned spins = 0;
            bool bad = false;
            for (;;) {
                bool ok = true;
                if (t >= 0) {
#pragma unroll
                    for (int c2 = 0; c2 < NCU; ++c2) gq[c2] = granule_load(grs, (unsigned)(par * NCU + c2) * MSG + col);
#pragma unroll
                    for (int c2 = 0; c2 < NCU; ++c2) ok = ok && granule_ok(gq[c2], epoch);
                }
                g1 = granule_load(grs, r1g + col);
                ok = ok && granule_ok(g1, epoch);
                if (__all(ok)) break;
                if (++spins > BASQ_GRANULE_SPIN_LIMIT) { bad = true; break; }   // never in a healthy run (wave-uniform)
                if (spins > 16) __builtin_amdgcn_s_sleep(2);
            }
            double tot = 0.0;
            if (t >= 0) {
#pragma unroll
                for (int c2 = 0; c2 < NCU; ++c2) tot += granule_value(gq[c2]);   // cluster order: the same sum everywhere
            }
            tot_l[col] = tot;
            r1_l[(size_t)par * MSG + col] = granule_value(g1);
            if (bad && lane == 0) abort_l = 1;
            __syncthreads();                                        // sums, row and verdict: work-group uniform from here
            aborted = abort_l != 0;
            if (aborted) break;                                     // no wave is left behind at a barrier
#pragma unroll
            for (int h = 0; h < NV / 2; ++h) {
                const d2_t v = *reinterpret_cast<const d2_t*>(tot_l + 2 * lane + 128 * h);
                accs[2 * h] = v.x;
                accs[2 * h + 1] = v.y;
            }
            ss2 = tot_l[NC - 1];
            if (lane == 63) accs[NV - 1] = 0.0;                     // (slot NC - 1 carried the scalar)
        }
        {
            const double* r1src = r1_l + (size_t)par * MSG;          // clusters: wave 0's copy of the published row
#pragma unroll
            for (int h = 0; h < NV / 2; ++h) {
                const d2_t v = *reinterpret_cast<const d2_t*>(r1src + 2 * lane + 128 * h);
                r1[2 * h] = v.x;
                r1[2 * h + 1] = v.y;
            }
            alphaH = r1src[GLOBAL ? NC - 1 : NC];
            if (GLOBAL && lane == 63) r1[NV - 1] = 0.0;
        }
        BASQ_NS_STAMP(t + 1, 4);
        if (t >= 0) {
            double tauq, uscale;
            householder_params_fast(alphaH, ss2, tauq, uscale);
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int c = BASQ_PAIRCOL(k);
                const double w_c = (c > t) ? (r1[k] + uscale * accs[k]) : 0.0;     // u^T A with u = [1, column * scale]
                wc[k] = w_c;
                rn[k] = r1[k] - tauq * w_c;                                          // row t+1 after H_t
            }
            kappa = tauq * uscale;
        } else {
#pragma unroll
            for (int k = 0; k < NV; ++k) rn[k] = r1[k];
        }
        {   // right reflector G_{t+1} from rn
            const int t1 = t + 1, tk = (t1 & 1) + 2 * (t1 >> 7), tl = (t1 & 127) >> 1;
            double ss = 0.0;
#pragma unroll
            for (int k = 0; k < NV; ++k)
                if (BASQ_PAIRCOL(k) > t1) ss += rn[k] * rn[k];
            BASQ_NS_STAMP(t + 1, 5);
            ss = wave_sum(ss);
            const double alpha = readlane_f64(pick_slot<NV>(rn, tk), tl);
            BASQ_NS_STAMP(t + 1, 6);
            double scale;
            householder_params_fast(alpha, ss, tau, scale);
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int c = BASQ_PAIRCOL(k);
                vr[k] = (c < t1) ? 0.0 : ((c == t1) ? 1.0 : rn[k] * scale);
            }
            if (gw == t1 % W) {
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    const int c = BASQ_PAIRCOL(k);
                    if (c < n) V[(size_t)t1 * n + c] = vr[k];
                }
                if (lane == 0) tau_g[t1] = tau;
            }
        }
        BASQ_NS_STAMP(t + 1, 3);
    }
    if (aborted && threadIdx.x == 0) {   // ANY member that gave up reports (its siblings run into the same limit, or are done)
        tau_g[0] = __builtin_nan("");   // poisons the null space: an elimination that ignores `info` still fails loudly
        if (info_g) info_g[0] = 2;      // status 2: a bounded spin expired (sibling work-groups not co-resident)
    }
}

template <int NV>
__global__ void __launch_bounds__(256) nullspace_apply_kernel(const double* __restrict__ V,
                                                              const double* __restrict__ tau, int m, int n,
                                                              double* __restrict__ PhiT) {
    // Reflector rows in flight.  V was written a moment ago by ONE work-group: for the other XCDs its lines come from
    // memory, not from their L2 (~1.2 us), i.e. four reductions' worth of latency is not enough -- 4 rows: 31.3 us,
    // 8: 28.1, 12: 25.7, 16: 25.6 (profiles/r02_m_nullspace_apply_rows_in_flight.txt)
    constexpr int PF = 12;
    const int lane = threadIdx.x & 63;
    const int c0 = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c0 >= n - m) return;                              // wave-uniform
    double y[NV], v[PF][NV], tv[PF];
#pragma unroll
    for (int k = 0; k < NV; ++k) y[k] = (lane + 64 * k == m + c0) ? 1.0 : 0.0;
#pragma unroll
    for (int p = 0; p < PF; ++p) {
        const int i = m - 1 - p;
        tv[p] = (i >= 0) ? tau[i] : 0.0;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int c = lane + 64 * k;
            v[p][k] = (i >= 0 && c < n) ? V[(size_t)i * n + c] : 0.0;
        }
    }
    for (int i0 = m - 1; i0 >= 0; i0 -= PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {                    // static ring slot p holds row i0 - p
            const int i = i0 - p;
            if (i < 0) break;                             // wave-uniform
            double dot = 0.0;
#pragma unroll
            for (int k = 0; k < NV; ++k) dot += v[p][k] * y[k];
            dot = wave_sum(dot);
            const double t = tv[p] * dot;
            const int inext = i - PF;                     // refill the slot behind the reduction
            tv[p] = (inext >= 0) ? tau[inext] : 0.0;
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int c = lane + 64 * k;
                y[k] -= t * v[p][k];
                v[p][k] = (inext >= 0 && c < n) ? V[(size_t)inext * n + c] : 0.0;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c = lane + 64 * k;
        if (c < n) PhiT[(size_t)c0 * n + c] = y[k];
    }
}

template <int NV, int NREG, int NW, bool ROWS_IN_LDS>
static int launch_bidiag(const double* X, int m, int n, double* V, double* tau, size_t lds, hipStream_t st) {
    auto kern = bidiag_reflectors_kernel<NV, NREG, NW, ROWS_IN_LDS>;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return BASQ_ELAUNCH;
    hipLaunchKernelGGL(kern, dim3(1), dim3(NW * 64), lds, st, X, m, n, V, tau);
    BASQ_CHECK_LAUNCH();
    return BASQ_OK;
}

// ------------------------------------------------------------------------------------------------
// Survivor re-weighting + order-preserving compaction (BASQ/_rchq.py:107-130).
// ------------------------------------------------------------------------------------------------
__global__ void reweight_compact_kernel(const double* __restrict__ cand, const double* __restrict__ mu,
                                        const long long* __restrict__ gid, const double* __restrict__ wx,
                                        long long Rl, long long off, long long n_full, int S, int kp,
                                        const int* __restrict__ keep_rank, const double* __restrict__ w_star,
                                        const double* __restrict__ tot, int n_keep, long long new_off,
                                        double* __restrict__ cand_out, double* __restrict__ mu_out,
                                        long long* __restrict__ gid_out, double* __restrict__ wx_out) {
#pragma clang fp contract(off)
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= Rl * kp) return;
    const long long p = t / kp;
    const int k = (int)(t - p * kp);
    const long long pg = off + p;
    int set;
    long long dst;
    if (pg < n_full) {
        const long long blk = pg / S;
        set = (int)(pg - blk * S);
        dst = blk * n_keep;
    } else {
        set = S - 1;
        dst = (n_full / S) * n_keep + (pg - n_full);
    }
    const int kr = keep_rank[set];
    if (kr < 0) return;
    if (pg < n_full) dst += kr;
    dst -= new_off;
    cand_out[dst * kp + k] = cand[t];
    if (k == 0) {
        const double scaled = mu[p] * w_star[kr];                          // :113-114 / :121-122
        mu_out[dst] = scaled / tot[set];
        gid_out[dst] = gid[p];
        if (wx) wx_out[dst] = wx[p];
    }
}

// Descriptor-driven form: the shard [off, off + Rl) of this rank, the block geometry and the number of kept sets are read
// from device memory (this round's and the next round's descriptors + the elimination's info word), the grid is sized
// for an upper bound of the shard.
__global__ void reweight_compact_geo_kernel(const double* __restrict__ cand, const double* __restrict__ mu,
                                            const long long* __restrict__ gid, const double* __restrict__ wx,
                                            const long long* __restrict__ geo, const long long* __restrict__ geo_next,
                                            const int* __restrict__ info, int S, int kp,
                                            const int* __restrict__ keep_rank, const double* __restrict__ w_star,
                                            const double* __restrict__ tot, long long out_rows, int expect_keep,
                                            double* __restrict__ cand_out, double* __restrict__ mu_out,
                                            long long* __restrict__ gid_out, double* __restrict__ wx_out) {
#pragma clang fp contract(off)
    const long long n_full = geo[1], off = geo[6], Rl = geo[7];
    const int n_keep = info[0];
    // The host sized the outputs for `expect_keep` kept sets before it knew the outcome: a round that violates that (failed
    // or short elimination, or the sticky flag of an earlier round) writes NOTHING -- the host repeats the rounds one
    // read-back at a time once it reads the flag (basq_round_next_i64 publishes it and an empty next round).
    if (geo[3] != 0 || info[1] != 0 || (expect_keep >= 0 && n_keep != expect_keep)) return;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= Rl * kp) return;
    const long long p = t / kp;
    const int k = (int)(t - p * kp);
    const long long pg = off + p;
    int set;
    long long dst;
    if (pg < n_full) {
        const long long blk = pg / S;
        set = (int)(pg - blk * S);
        dst = blk * n_keep;
    } else {
        set = S - 1;
        dst = (n_full / S) * n_keep + (pg - n_full);
    }
    const int kr = keep_rank[set];
    if (kr < 0) return;
    if (pg < n_full) dst += kr;
    dst -= geo_next[6];                                                    // this rank's new offset
    if (dst < 0 || dst >= out_rows) return;                                // never outside the caller's buffers
    cand_out[dst * kp + k] = cand[t];
    if (k == 0) {
        const double scaled = mu[p] * w_star[kr];                          // :113-114 / :121-122
        mu_out[dst] = scaled / tot[set];
        gid_out[dst] = gid[p];
        if (wx) wx_out[dst] = wx[p];
    }
}

// Next round's descriptor from this round's outcome (BASQ/_rchq.py:107-130 in closed form, basq_amd/_partition.py):
//   R' = nb * n_keep + (n_tail if set S-1 survived);  class_mode > 0: a fresh evaluation with that many residue classes
//   (regular region = the largest multiple of class_mode blocks), -1: the classes are inherited (regular region halves),
//   0: no classes.  Sticky violation flag: the elimination failed (status) or did not keep exactly half of the sets
//   while the host had already enqueued a regrouping that relies on it -- the host then repeats the batch round by round.
__device__ __forceinline__ void round_next_body(const int lane, const long long* __restrict__ gp,
                                                const int* __restrict__ info, const int* __restrict__ keep_rank, int S,
                                                int class_mode, int expect_half, long long* __restrict__ gn) {
    // one wave: the lanes share the two prefix counts of the shard's new offset / end (survivors_before of _partition.py)
    const long long R = gp[0], n_full = gp[1], off = gp[6], Rl = gp[7];
    const long long nb = n_full / S, n_tail = R - n_full;
    const int n_keep = info[0], status = info[1];
    long long viol = gp[3];
    if (status != 0 || (expect_half && 2 * n_keep != S)) viol = 1;
    const bool last_kept = keep_rank[S - 1] >= 0;
    long long before[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const long long P = e ? (off + Rl) : off;
        if (P <= n_full) {
            const long long b = P / S;
            const int sidx = (int)(P - b * S);
            int cnt = 0;
            for (int j = lane; j < sidx; j += 64) cnt += (keep_rank[j] >= 0) ? 1 : 0;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
            before[e] = b * n_keep + cnt;
        } else {
            before[e] = nb * n_keep + (last_kept ? (P - n_full) : 0);
        }
    }
    if (lane != 0) return;
    // after a violation every later descriptor-driven launch of the batch sees an EMPTY round (the buffers the host sized
    // for the expected survivor counts are never overrun); the flag tells the host to repeat the rounds one by one
    const long long Rn = viol ? 0 : nb * n_keep + (last_kept ? n_tail : 0);
    const long long nbn = Rn / S;
    long long reg_blocks = 0;
    if (class_mode > 0) reg_blocks = (nbn / class_mode) * class_mode;
    else if (class_mode < 0) reg_blocks = (gp[2] / S) / 2;
    gn[0] = Rn;
    gn[1] = nbn * S;
    gn[2] = reg_blocks * S;
    gn[3] = viol;
    gn[4] = nbn;
    gn[5] = Rn - nbn * S;
    gn[6] = viol ? 0 : before[0];
    gn[7] = viol ? 0 : (before[1] - before[0]);
}

__global__ void round_next_kernel(const long long* __restrict__ gp, const int* __restrict__ info,
                                  const int* __restrict__ keep_rank, int S, int class_mode, int expect_half,
                                  long long* __restrict__ gn) {
    if (blockIdx.x == 0 && threadIdx.x < 64) round_next_body(threadIdx.x, gp, info, keep_rank, S, class_mode, expect_half, gn);
}

// ------------------------------------------------------------------------------------------------
// Small dense Cholesky + triangular inverse, one work-group, in place in global memory (L2-resident):
// G (SPD, q x q) -> L in the lower triangle;  W = L^{-T} (upper triangular), so that for X with
// X^T X = G the matrix Q = X W has orthonormal columns (CholeskyQR step of the randomised SVD).
// info[0] = 0, or j+1 if pivot j fell below rel_tol * max diag (caller falls back to Householder QR).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) chol_inv_kernel(double* __restrict__ G, int q, double* __restrict__ W,
                                                        int* __restrict__ info, double rel_tol) {
    __shared__ double colj[1024];
    __shared__ double red[16];
    __shared__ double s_dmax;
    __shared__ int s_bad;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double dm = (tid < q) ? G[(long long)tid * q + tid] : 0.0;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) dm = fmax(dm, __shfl_xor(dm, o, 64));
    if (lane == 0) red[wv] = dm;
    if (tid == 0) s_bad = 0;
    __syncthreads();
    if (tid == 0) {
        double v = red[0];
        for (int w = 1; w < 16; ++w) v = fmax(v, red[w]);
        s_dmax = v;
    }
    for (long long e = tid; e < (long long)q * q; e += 1024) W[e] = 0.0;
    __syncthreads();
    const double floor_ = rel_tol * s_dmax;
    for (int j = 0; j < q; ++j) {
        if (tid == 0) {
            const double d = G[(long long)j * q + j];
            if (!(d > floor_)) s_bad = j + 1;
            else G[(long long)j * q + j] = sqrt(d);
        }
        __syncthreads();
        if (s_bad) break;                                   // uniform
        const double piv = G[(long long)j * q + j];
        for (int i = j + 1 + tid; i < q; i += 1024) {
            const double v = G[(long long)i * q + j] / piv;
            G[(long long)i * q + j] = v;
            colj[i] = v;
        }
        __syncthreads();
        const int n = q - j - 1;
        for (int idx = tid; idx < n * n; idx += 1024) {
            const int a = idx / n, b = idx - a * n;
            if (b <= a) {
                const int i = j + 1 + a, k = j + 1 + b;
                G[(long long)i * q + k] -= colj[i] * colj[k];
            }
        }
        __syncthreads();
    }
    if (tid == 0) info[0] = s_bad;
    if (s_bad || tid >= q) return;
    // column c of Y = L^{-1} by forward substitution; stored as row c of W (= Y^T)
    const int c = tid;
    double* wrow = W + (long long)c * q;
    for (int i = c; i < q; ++i) {
        const double* lrow = G + (long long)i * q;
        double acc = (i == c) ? 1.0 : 0.0;
        for (int k = c; k < i; ++k) acc -= lrow[k] * wrow[k];
        wrow[i] = acc / lrow[i];
    }
}

// LDS-resident form of chol_inv_kernel for q*q doubles <= ~150 KB (q <= 136): the factor lives in LDS with an
// odd leading dimension.  Cholesky: right-looking, column j scaled by a reciprocal square root (no sqrt + divide
// chain), 2 barriers per column.  Inverse: with BLOCKED != 0 (a second q x q square fits in LDS, q <= 100)
// Y = L^{-1} is assembled from the inverses of four diagonal blocks (one thread per column, chains of
// (q/4)^2/2 steps instead of q^2/2) and two levels of products Y_CA = -Y_CC (L_CA Y_AA) done by all threads;
// otherwise one thread per column runs the whole forward substitution.
__global__ void __launch_bounds__(1024) chol_inv_lds_kernel(double* __restrict__ G, int q, double* __restrict__ W,
                                                            int* __restrict__ info, double rel_tol, int blocked) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int ld = q | 1;
    double* Ls = sm;                  // [q][ld]
    double* colj = sm + (size_t)q * ld;            // [q rounded up to even]: column j, then the reciprocal diagonal
    double* Ys = colj + ((q + 1) & ~1);            // [q][ld]  (only when blocked)
    __shared__ double red[16];
    __shared__ double s_dmax;
    __shared__ int s_bad;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nt = blockDim.x, nwv = nt >> 6, rows_pp = nt >> 7;   // 256..1024 threads (BASQ_CHOL_THREADS)
    for (int e = tid; e < q * q; e += nt) {
        const int i = e / q, k = e - i * q;
        Ls[i * ld + k] = G[e];
        W[e] = 0.0;
    }
    if (tid == 0) s_bad = 0;
    __syncthreads();
    double dm = (tid < q) ? Ls[tid * ld + tid] : 0.0;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) dm = fmax(dm, __shfl_xor(dm, o, 64));
    if (lane == 0) red[wv] = dm;
    __syncthreads();
    if (tid == 0) {
        double v = red[0];
        for (int w = 1; w < nwv; ++w) v = fmax(v, red[w]);
        s_dmax = v;
    }
    __syncthreads();
    const double floor_ = rel_tol * s_dmax;
    BASQ_NS_STAMP(0, 0);
    for (int j = 0; j < q; ++j) {
        const double d = Ls[j * ld + j];
        if (!(d > floor_)) {                                  // uniform: every thread reads the same pivot
            if (tid == 0) s_bad = j + 1;
            break;
        }
        double rpiv = 0.0;
        if (tid < q - j) {                                    // only the waves that hold column j do the pivot math
            rpiv = rsqrt_nr(d);
            for (int i = j + tid; i < q; i += nt) {
                if (i == j) continue;
                const double v = Ls[i * ld + j] * rpiv;
                colj[i] = v;
                Ls[i * ld + j] = v;
            }
        }
        __syncthreads();
        if (tid == 0) {                                       // after the barrier: every thread has read d by now
            Ls[j * ld + j] = d * rpiv;
            colj[j] = rpiv;                                   // reciprocal diagonal, used by the inverse
        }
        // trailing update, lower triangle: nt/128 rows x 128 columns per pass (no index division)
        for (int a = j + 1 + (tid >> 7); a < q; a += rows_pp) {
            for (int b = j + 1 + (tid & 127); b <= a; b += 128) Ls[a * ld + b] -= colj[a] * colj[b];
        }
        __syncthreads();
    }
    __syncthreads();
    BASQ_NS_STAMP(0, 1);
    const int bad = s_bad;
    if (tid == 0) info[0] = bad;
    // write L back (lower triangle incl. diagonal; the strict upper triangle of G is left untouched)
    for (int e = tid; e < q * q; e += nt) {
        const int i = e / q, k = e - i * q;
        if (k <= i) G[e] = Ls[i * ld + k];
    }
    BASQ_NS_STAMP(0, 2);
    if (bad) return;                                          // uniform
    if (blocked) {
        // ---- blocked inverse: Y = L^{-1} in Ys (lower triangle); products staged transposed in Ys' upper triangle
        int bnd[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) bnd[k] = (k * q) / 4;
        // (1) diagonal blocks: thread c inverts its column inside its block
        for (int c = tid; c < q; c += nt) {
            int hi = q;
#pragma unroll
            for (int k = 4; k >= 1; --k) if (c < bnd[k]) hi = bnd[k];
            const double ycc = colj[c];
            Ys[c * ld + c] = ycc;
            for (int i = c + 1; i < hi; ++i) {
                const double* lrow = Ls + i * ld;
                double a0 = -lrow[c] * ycc, a1 = 0.0;
                int k = c + 1;
                for (; k + 1 < i; k += 2) {
                    a0 = __builtin_fma(-lrow[k], Ys[k * ld + c], a0);
                    a1 = __builtin_fma(-lrow[k + 1], Ys[(k + 1) * ld + c], a1);
                }
                if (k < i) a0 = __builtin_fma(-lrow[k], Ys[k * ld + c], a0);
                Ys[i * ld + c] = (a0 + a1) * colj[i];
            }
        }
        __syncthreads();
        // (2) merge levels: pairs (0,1), (2,3), then (01, 23).  For rows r in C = [c0, c1), columns c in A = [a0, c0):
        //     T[r][c] = sum_{k=c}^{c0-1} L[r][k] Y[k][c]  -> staged at Ys[c][r];   Y[r][c] = -sum_{k=c0}^{r} Y[r][k] T[k][c]
        for (int level = 0; level < 2; ++level) {
            const int npair = level == 0 ? 2 : 1;
            for (int phase = 0; phase < 2; ++phase) {
                for (int pr = 0; pr < npair; ++pr) {
                    const int a0 = level == 0 ? bnd[2 * pr] : bnd[0];
                    const int c0 = level == 0 ? bnd[2 * pr + 1] : bnd[2];
                    const int c1 = level == 0 ? bnd[2 * pr + 2] : bnd[4];
                    const int nA = c0 - a0, nC = c1 - c0;
                    if (nA <= 0 || nC <= 0) continue;
                    for (int o = tid; o < nA * nC; o += nt) {
                        const int rr = o / nA, cc = o - rr * nA;        // consecutive threads -> consecutive columns
                        const int r = c0 + rr, c = a0 + cc;
                        double acc0 = 0.0, acc1 = 0.0;
                        if (phase == 0) {
                            const double* lrow = Ls + r * ld;
                            int k = c;
                            for (; k + 1 < c0; k += 2) {
                                acc0 = __builtin_fma(lrow[k], Ys[k * ld + c], acc0);
                                acc1 = __builtin_fma(lrow[k + 1], Ys[(k + 1) * ld + c], acc1);
                            }
                            if (k < c0) acc0 = __builtin_fma(lrow[k], Ys[k * ld + c], acc0);
                            Ys[c * ld + r] = acc0 + acc1;               // staged transposed (strict upper triangle)
                        } else {
                            const double* yrow = Ys + r * ld;
                            const double* trow = Ys + c * ld;
                            int k = c0;
                            for (; k + 1 <= r; k += 2) {
                                acc0 = __builtin_fma(yrow[k], trow[k], acc0);
                                acc1 = __builtin_fma(yrow[k + 1], trow[k + 1], acc1);
                            }
                            if (k <= r) acc0 = __builtin_fma(yrow[k], trow[k], acc0);
                            Ys[r * ld + c] = -(acc0 + acc1);
                        }
                    }
                }
                __syncthreads();
            }
        }
        // W = L^{-T}: W[c][i] = Y[i][c], i >= c (W was zeroed above)
        for (int e = tid; e < q * q; e += nt) {
            const int i = e / q, c = e - i * q;
            if (c <= i) W[(long long)c * q + i] = Ys[i * ld + c];
        }
        BASQ_NS_STAMP(0, 3);
        return;
    }
    // column c of Y = L^{-1} by forward substitution, kept in the (now free) upper triangle of Ls:
    // Y[i][c] (i >= c) is stored at Ls[c][i] for i > c (strictly upper), and its diagonal in a register.
    for (int c = tid; c < q; c += nt) {
    double* wrow = W + (long long)c * q;
    const double* yrow = Ls + c * ld;
    const double ycc = colj[c];
    wrow[c] = ycc;
    for (int i = c + 1; i < q; ++i) {
        const double* lrow = Ls + i * ld;
        double a0 = -lrow[c] * ycc, a1 = 0.0, a2 = 0.0, a3 = 0.0;     // four independent chains (latency-bound loop)
        int k = c + 1;
        for (; k + 3 < i; k += 4) {
            a0 = __builtin_fma(-lrow[k], yrow[k], a0);
            a1 = __builtin_fma(-lrow[k + 1], yrow[k + 1], a1);
            a2 = __builtin_fma(-lrow[k + 2], yrow[k + 2], a2);
            a3 = __builtin_fma(-lrow[k + 3], yrow[k + 3], a3);
        }
        for (; k < i; ++k) a0 = __builtin_fma(-lrow[k], yrow[k], a0);
        const double y = ((a0 + a1) + (a2 + a3)) * colj[i];
        Ls[c * ld + i] = y;                                   // row c, column i > c: strictly upper, owned by thread c
        wrow[i] = y;
    }
    }
    BASQ_NS_STAMP(0, 3);
}

// Factor-only form for Gram matrices whose square does not fit in LDS but whose lower triangle does (142 < q <= 200,
// e.g. q = 199 for batches of 200): packed row-major lower triangle L(i,j) at i(i+1)/2 + j, same right-looking
// steps as above.  The caller obtains W = L^{-T} from a library triangular solve (plumbing, like the GEMMs).
__global__ void __launch_bounds__(1024) chol_packed_lds_kernel(double* __restrict__ G, int q, int* __restrict__ info,
                                                               double rel_tol) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* Lp = sm;                                     // [q (q + 1) / 2]
    double* colj = sm + (size_t)q * (q + 1) / 2;         // [q]
    __shared__ double red[16];
    __shared__ double s_dmax;
    __shared__ int s_bad;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
#define BASQ_TRI(i, j) Lp[(size_t)(i) * ((i) + 1) / 2 + (j)]
    for (int i = tid >> 7; i < q; i += 8)
        for (int k = tid & 127; k <= i; k += 128) BASQ_TRI(i, k) = G[(size_t)i * q + k];
    if (tid == 0) s_bad = 0;
    __syncthreads();
    double dm = (tid < q) ? BASQ_TRI(tid, tid) : 0.0;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) dm = fmax(dm, __shfl_xor(dm, o, 64));
    if (lane == 0) red[wv] = dm;
    __syncthreads();
    if (tid == 0) {
        double v = red[0];
        for (int w = 1; w < 16; ++w) v = fmax(v, red[w]);
        s_dmax = v;
    }
    __syncthreads();
    const double floor_ = rel_tol * s_dmax;
    for (int j = 0; j < q; ++j) {
        const double d = BASQ_TRI(j, j);
        if (!(d > floor_)) {                                  // uniform: every thread reads the same pivot
            if (tid == 0) s_bad = j + 1;
            break;
        }
        double rpiv = 0.0;
        if (tid < q - j) {
            rpiv = rsqrt_nr(d);
            if (tid > 0) {
                const int i = j + tid;
                const double v = BASQ_TRI(i, j) * rpiv;
                colj[i] = v;
                BASQ_TRI(i, j) = v;
            }
        }
        __syncthreads();
        if (tid == 0) BASQ_TRI(j, j) = d * rpiv;              // after the barrier: every thread has read d by now
        for (int a = j + 1 + (tid >> 7); a < q; a += 8) {
            const double ca = colj[a];
            for (int b = j + 1 + (tid & 127); b <= a; b += 128) BASQ_TRI(a, b) -= ca * colj[b];
        }
        __syncthreads();
    }
    __syncthreads();
    if (tid == 0) info[0] = s_bad;
    for (int i = tid >> 7; i < q; i += 8)
        for (int k = tid & 127; k <= i; k += 128) G[(size_t)i * q + k] = BASQ_TRI(i, k);
#undef BASQ_TRI
}

// ------------------------------------------------------------------------------------------------
// CholeskyQR building blocks of the range finder, second generation: a PANEL Cholesky and a row-parallel triangular
// solve, so that no inverse is formed and the factorisation costs q/8 synchronised steps instead of q.
//
// chol_factor_panel_kernel: one work-group, the lower triangle packed in LDS (q <= 200).  Per panel of 8 columns:
//   F1  every thread factors the 8 x 8 diagonal block redundantly in registers (broadcast LDS reads; nothing to hand
//       over, the pivot test is uniform by construction);
//   F2  one thread per row below the block: its 8 panel entries by forward substitution against the block;
//   F3  the trailing triangle in 4 x 4 tiles: A[i][k] -= sum_c L[i][c] L[k][c].
// Two barriers per panel.
// trsm_rows_kernel: Q = X L^-T for a tall X [rows, q]: 64 rows per work-group, their q entries in LDS; per panel the
//   512 threads (row, panel column) subtract the contribution of the finished columns (dot products over LDS rows and
//   L rows that are uniform per wave), then one thread per row solves its 8 x 8 block.  Replaces W = L^-T + a GEMM.
// ------------------------------------------------------------------------------------------------
#define BASQ_CHOL_NB 8
template <int NTHR>
__global__ void __launch_bounds__(NTHR) chol_factor_panel_kernel(double* __restrict__ G, int q, int* __restrict__ info,
                                                                 double rel_tol) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* Lp = sm;                                     // [q (q + 1) / 2] packed rows
    __shared__ double red[16];
    __shared__ double s_dmax;
    constexpr int NB = BASQ_CHOL_NB;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
#define BASQ_TRI(i, j) Lp[(size_t)(i) * ((i) + 1) / 2 + (j)]
    for (int i = tid >> 7; i < q; i += NTHR / 128)
        for (int k = tid & 127; k <= i; k += 128) BASQ_TRI(i, k) = G[(size_t)i * q + k];
    __syncthreads();
    double dm = 0.0;
    for (int i = tid; i < q; i += NTHR) dm = fmax(dm, BASQ_TRI(i, i));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) dm = fmax(dm, __shfl_xor(dm, o, 64));
    if (lane == 0) red[wv] = dm;
    __syncthreads();
    if (tid == 0) {
        double v = red[0];
        for (int w = 1; w < NTHR / 64; ++w) v = fmax(v, red[w]);
        s_dmax = v;
    }
    __syncthreads();
    const double floor_ = rel_tol * s_dmax;
    int bad = 0;
    for (int j0 = 0; j0 < q; j0 += NB) {
        const int nb = (q - j0 < NB) ? (q - j0) : NB;
        // ---- F1: diagonal block, redundantly in every thread ----
        double D[NB][NB], rd[NB];
#pragma unroll
        for (int r = 0; r < NB; ++r)
#pragma unroll
            for (int c = 0; c <= r; ++c) D[r][c] = (r < nb) ? BASQ_TRI(j0 + r, j0 + c) : ((r == c) ? 1.0 : 0.0);
#pragma unroll
        for (int c = 0; c < NB; ++c) {
            const double d = D[c][c];
            if (c < nb && !(d > floor_) && bad == 0) bad = j0 + c + 1;       // uniform: every thread holds the same block
            const double r = rsqrt_nr(bad ? 1.0 : d);
            rd[c] = r;
            D[c][c] = d * r;
#pragma unroll
            for (int i = c + 1; i < NB; ++i) D[i][c] *= r;
#pragma unroll
            for (int i = c + 1; i < NB; ++i)
#pragma unroll
                for (int k = c + 1; k <= i; ++k) D[i][k] -= D[i][c] * D[k][c];
        }
        if (bad) break;                                                     // uniform
        // ---- F2: rows below the block ----
        const int R = q - j0 - nb;
        if (tid < R) {
            const int i = j0 + nb + tid;
            double y[NB];
#pragma unroll
            for (int c = 0; c < NB; ++c) y[c] = (c < nb) ? BASQ_TRI(i, j0 + c) : 0.0;
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                double v = y[c];
#pragma unroll
                for (int k = 0; k < c; ++k) v -= y[k] * D[c][k];
                y[c] = v * rd[c];
            }
#pragma unroll
            for (int c = 0; c < NB; ++c)
                if (c < nb) BASQ_TRI(i, j0 + c) = y[c];
        }
        __syncthreads();
        // the factored block goes back only now: before the barrier a slower wave may still be READING the unfactored
        // block in F1 (F3 below touches neither the block nor these rows' panel entries)
        if (tid == NTHR - 1) {
#pragma unroll
            for (int r = 0; r < NB; ++r)
#pragma unroll
                for (int c = 0; c <= r; ++c)
                    if (r < nb) BASQ_TRI(j0 + r, j0 + c) = D[r][c];
        }
        // ---- F3: trailing triangle, 4 x 4 tiles ----
        const int nt = (R + 3) >> 2, ntiles = nt * (nt + 1) / 2;
        for (int tile = tid; tile < ntiles; tile += NTHR) {
            int ti = (int)((__builtin_sqrtf(8.0f * (float)tile + 1.0f) - 1.0f) * 0.5f);
            while (ti * (ti + 1) / 2 > tile) --ti;
            while ((ti + 1) * (ti + 2) / 2 <= tile) ++ti;
            const int tk = tile - ti * (ti + 1) / 2;
            const int i0 = j0 + nb + 4 * ti, k0 = j0 + nb + 4 * tk;
            double acc[4][4];
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y2 = 0; y2 < 4; ++y2) acc[x][y2] = 0.0;
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                double li[4], lk[4];
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    const int ii = (i0 + x < q) ? (i0 + x) : (q - 1), kk2 = (k0 + x < q) ? (k0 + x) : (q - 1);
                    li[x] = (c < nb) ? BASQ_TRI(ii, j0 + c) : 0.0;
                    lk[x] = (c < nb) ? BASQ_TRI(kk2, j0 + c) : 0.0;
                }
#pragma unroll
                for (int x = 0; x < 4; ++x)
#pragma unroll
                    for (int y2 = 0; y2 < 4; ++y2) acc[x][y2] = __builtin_fma(li[x], lk[y2], acc[x][y2]);
            }
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y2 = 0; y2 < 4; ++y2)
                    if (i0 + x < q && k0 + y2 <= i0 + x) BASQ_TRI(i0 + x, k0 + y2) -= acc[x][y2];
        }
        __syncthreads();
    }
    if (tid == 0) info[0] = bad;
    for (int i = tid >> 7; i < q; i += NTHR / 128)
        for (int k = tid & 127; k <= i; k += 128) G[(size_t)i * q + k] = BASQ_TRI(i, k);
#undef BASQ_TRI
}

// LSH: the factor is copied into LDS first (one coalesced pass).  L was written a moment ago by ONE work-group, so for the
// other XCDs its lines come from memory (~1.2 us per dependent access), and every panel needs new rows of it twice: with L
// in global memory the 13 panels of q = 99 cost 60 us, nearly all of it those round trips.
template <bool LSH>
__global__ void __launch_bounds__(512) trsm_rows_kernel(const double* __restrict__ X, long long ldx, long long rows, int q,
                                                        const double* __restrict__ L, double* __restrict__ Qo,
                                                        long long ldq) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    constexpr int NB = BASQ_CHOL_NB;
    const int ld = q | 1;                                 // odd leading dimension: lanes (= rows) hit distinct banks
    double* Y = sm;                                       // [64][ld]
    double* Lsh = sm + 64 * ld;                           // [q][q] (LSH only)
    const int tid = threadIdx.x;
    const long long r0 = (long long)blockIdx.x * 64;
    const int nr = (rows - r0 < 64) ? (int)(rows - r0) : 64;
    if (LSH)
        for (int e = tid; e < q * q; e += 512) Lsh[e] = L[e];
    for (int e = tid; e < 64 * q; e += 512) {
        const int r = e / q, c = e - r * q;
        Y[r * ld + c] = (r < nr) ? X[(r0 + r) * ldx + c] : 0.0;
    }
    __syncthreads();
    const int r = tid & 63, c = tid >> 6;                 // wave = panel column c (uniform), lane = row
    auto panels = [&](const double* Lb) {
        for (int j0 = 0; j0 < q; j0 += NB) {
            const int nb = (q - j0 < NB) ? (q - j0) : NB;
            if (c < nb) {
                // s = y[r][j0 + c] - sum_{k < j0} y[r][k] L[j0 + c][k]   (L row uniform per wave; four chains)
                const double* lrow = Lb + (size_t)(j0 + c) * q;
                const double* yrow = Y + r * ld;
                double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
                int k = 0;
                for (; k + 3 < j0; k += 4) {
                    s0 = __builtin_fma(yrow[k], lrow[k], s0);
                    s1 = __builtin_fma(yrow[k + 1], lrow[k + 1], s1);
                    s2 = __builtin_fma(yrow[k + 2], lrow[k + 2], s2);
                    s3 = __builtin_fma(yrow[k + 3], lrow[k + 3], s3);
                }
                for (; k < j0; ++k) s0 = __builtin_fma(yrow[k], lrow[k], s0);
                Y[r * ld + j0 + c] -= (s0 + s1) + (s2 + s3);
            }
            __syncthreads();
            if (tid < 64) {                                   // one thread per row: the 8 x 8 block by forward substitution
                double y[NB];
#pragma unroll
                for (int cc = 0; cc < NB; ++cc) y[cc] = (cc < nb) ? Y[tid * ld + j0 + cc] : 0.0;
#pragma unroll
                for (int cc = 0; cc < NB; ++cc) {
                    if (cc < nb) {
                        const double* lrow = Lb + (size_t)(j0 + cc) * q + j0;
                        double v = y[cc];
#pragma unroll
                        for (int k = 0; k < cc; ++k) v -= y[k] * lrow[k];
                        y[cc] = v / lrow[cc];
                    }
                }
#pragma unroll
                for (int cc = 0; cc < NB; ++cc)
                    if (cc < nb) Y[tid * ld + j0 + cc] = y[cc];
            }
            __syncthreads();
        }
    };
    if (LSH) panels(Lsh);
    else panels(L);
    for (int e = tid; e < 64 * q; e += 512) {
        const int rr = e / q, cc = e - rr * q;
        if (rr < nr) Qo[(r0 + rr) * ldq + cc] = Y[rr * ld + cc];
    }
}

// ------------------------------------------------------------------------------------------------
// CholeskyQR in ONE launch: work-group 0 factors G = X^T X panel by panel (chol_factor_panel_kernel's steps), the others
// solve Q = X L^-T for 64 rows each (trsm_rows_kernel's steps) and start on panel p as soon as column panel p of L exists,
// instead of after the whole factor: 68 + 46 us (+ a launch) -> ~80 us at q = 99.
// Hand-over (MI355X_MICROARCH.md, visibility, "sc1 payload + drained flag"): the LAST wave of the factor work-group stores
// column panel p (rows j0.., 8 columns: final after F1/F2) write-through into G -- which is the kernel's output anyway --,
// drains its own stores and raises info[1] to p + 1; meanwhile the other seven waves run the trailing update.  A solving
// work-group polls that word with ONE lane (bounded), then copies the panel into its LDS image of L with L1-bypassing loads.
// Same arithmetic, same order as the two separate kernels: identical bits.
// ------------------------------------------------------------------------------------------------
#define BASQ_CHOLQR_ABORT 0x40000000u
template <bool FULL>   // FULL: a solver keeps an image of all of L in LDS (q <= 112); otherwise the 8 rows of the current panel
__global__ void __launch_bounds__(512) cholqr_fused_kernel(double* __restrict__ G, int q, int* __restrict__ info, double rel_tol,
                                                           const double* __restrict__ X, long long ldx, long long rows,
                                                           double* __restrict__ Qo, long long ldq) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    constexpr int NB = BASQ_CHOL_NB, NTHR = 512, NF3 = NTHR - 64;   // the last wave publishes while the others update
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    basq_gu32* progress = (basq_gu32*)(info + 1);
    if (blockIdx.x == 0) {
        // ---------------- factor ----------------
        double* Lp = sm;                                     // [q (q + 1) / 2] packed rows
        __shared__ double red[16];
        __shared__ double s_dmax;
#define BASQ_TRI(i, j) Lp[(size_t)(i) * ((i) + 1) / 2 + (j)]
        for (int i = tid >> 7; i < q; i += NTHR / 128)
            for (int k = tid & 127; k <= i; k += 128) BASQ_TRI(i, k) = G[(size_t)i * q + k];
        __syncthreads();
        double dm = 0.0;
        for (int i = tid; i < q; i += NTHR) dm = fmax(dm, BASQ_TRI(i, i));
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) dm = fmax(dm, __shfl_xor(dm, o, 64));
        if (lane == 0) red[wv] = dm;
        __syncthreads();
        if (tid == 0) {
            double v = red[0];
            for (int w = 1; w < NTHR / 64; ++w) v = fmax(v, red[w]);
            s_dmax = v;
        }
        __syncthreads();
        const double floor_ = rel_tol * s_dmax;
        int bad = 0, panel = 0;
        for (int j0 = 0; j0 < q; j0 += NB, ++panel) {
            const int nb = (q - j0 < NB) ? (q - j0) : NB;
            double D[NB][NB], rd[NB];
#pragma unroll
            for (int r = 0; r < NB; ++r)
#pragma unroll
                for (int c = 0; c <= r; ++c) D[r][c] = (r < nb) ? BASQ_TRI(j0 + r, j0 + c) : ((r == c) ? 1.0 : 0.0);
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                const double d = D[c][c];
                if (c < nb && !(d > floor_) && bad == 0) bad = j0 + c + 1;       // uniform: every thread holds the same block
                const double r = rsqrt_nr(bad ? 1.0 : d);
                rd[c] = r;
                D[c][c] = d * r;
#pragma unroll
                for (int i = c + 1; i < NB; ++i) D[i][c] *= r;
#pragma unroll
                for (int i = c + 1; i < NB; ++i)
#pragma unroll
                    for (int k = c + 1; k <= i; ++k) D[i][k] -= D[i][c] * D[k][c];
            }
            if (bad) break;                                                     // uniform
            const int R = q - j0 - nb;
            if (tid < R) {
                const int i = j0 + nb + tid;
                double y[NB];
#pragma unroll
                for (int c = 0; c < NB; ++c) y[c] = (c < nb) ? BASQ_TRI(i, j0 + c) : 0.0;
#pragma unroll
                for (int c = 0; c < NB; ++c) {
                    double v = y[c];
#pragma unroll
                    for (int k = 0; k < c; ++k) v -= y[k] * D[c][k];
                    y[c] = v * rd[c];
                }
#pragma unroll
                for (int c = 0; c < NB; ++c)
                    if (c < nb) BASQ_TRI(i, j0 + c) = y[c];
            }
            __syncthreads();
            if (wv == NTHR / 64 - 1) {
                // the factored block goes back (its last lane), then this wave publishes column panel `panel`
                if (tid == NTHR - 1) {
#pragma unroll
                    for (int r = 0; r < NB; ++r)
#pragma unroll
                        for (int c = 0; c <= r; ++c)
                            if (r < nb) BASQ_TRI(j0 + r, j0 + c) = D[r][c];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // (no instruction) the LDS writes stay above the reads
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                for (int e = lane; e < (q - j0) * NB; e += 64) {
                    const int i = j0 + e / NB, c = e % NB;
                    if (c < nb && j0 + c <= i)
                        __hip_atomic_store((basq_gu64*)(G + (size_t)i * q + j0 + c),
                                           (unsigned long long)__double_as_longlong(BASQ_TRI(i, j0 + c)), BASQ_RLX_AGENT);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's stores have left the CU
                if (lane == 0) __hip_atomic_store(progress, (unsigned)(panel + 1), BASQ_RLX_AGENT);
            } else {
                // trailing triangle, 4 x 4 tiles, on the other seven waves
                const int nt = (R + 3) >> 2, ntiles = nt * (nt + 1) / 2;
                for (int tile = tid; tile < ntiles; tile += NF3) {
                    int ti = (int)((__builtin_sqrtf(8.0f * (float)tile + 1.0f) - 1.0f) * 0.5f);
                    while (ti * (ti + 1) / 2 > tile) --ti;
                    while ((ti + 1) * (ti + 2) / 2 <= tile) ++ti;
                    const int tk = tile - ti * (ti + 1) / 2;
                    const int i0 = j0 + nb + 4 * ti, k0 = j0 + nb + 4 * tk;
                    double acc[4][4];
#pragma unroll
                    for (int x = 0; x < 4; ++x)
#pragma unroll
                        for (int y2 = 0; y2 < 4; ++y2) acc[x][y2] = 0.0;
#pragma unroll
                    for (int c = 0; c < NB; ++c) {
                        double li[4], lk[4];
#pragma unroll
                        for (int x = 0; x < 4; ++x) {
                            const int ii = (i0 + x < q) ? (i0 + x) : (q - 1), kk2 = (k0 + x < q) ? (k0 + x) : (q - 1);
                            li[x] = (c < nb) ? BASQ_TRI(ii, j0 + c) : 0.0;
                            lk[x] = (c < nb) ? BASQ_TRI(kk2, j0 + c) : 0.0;
                        }
#pragma unroll
                        for (int x = 0; x < 4; ++x)
#pragma unroll
                            for (int y2 = 0; y2 < 4; ++y2) acc[x][y2] = __builtin_fma(li[x], lk[y2], acc[x][y2]);
                    }
#pragma unroll
                    for (int x = 0; x < 4; ++x)
#pragma unroll
                        for (int y2 = 0; y2 < 4; ++y2)
                            if (i0 + x < q && k0 + y2 <= i0 + x) BASQ_TRI(i0 + x, k0 + y2) -= acc[x][y2];
                }
            }
            __syncthreads();
        }
        if (tid == 0) {
            info[0] = bad;
            if (bad) __hip_atomic_store(progress, BASQ_CHOLQR_ABORT, BASQ_RLX_AGENT);   // the solvers stop waiting
        }
#undef BASQ_TRI
        return;
    }
    // ---------------- solve: 64 rows of X per work-group ----------------
    const int ld = q | 1;
    double* Y = sm;                                       // [64][ld]
    double* Lsh = sm + 64 * ld;                           // FULL: [q][q]; else [NB][q] = rows j0.. of the current panel
    __shared__ unsigned s_seen;
    const long long r0 = (long long)(blockIdx.x - 1) * 64;
    const int nr = (rows - r0 < 64) ? (int)(rows - r0) : 64;
    for (int e = tid; e < 64 * q; e += 512) {
        const int r = e / q, c = e - r * q;
        Y[r * ld + c] = (r < nr) ? X[(r0 + r) * ldx + c] : 0.0;
    }
    const int r = tid & 63, c = tid >> 6;                 // wave = panel column c (uniform), lane = row
    int panel = 0;
    for (int j0 = 0; j0 < q; j0 += NB, ++panel) {
        const int nb = (q - j0 < NB) ? (q - j0) : NB;
        if (tid == 0) {                                   // ONE lane polls the progress word
            unsigned seen, spins = 0;
            for (;;) {
                seen = __hip_atomic_load(progress, BASQ_RLX_AGENT);
                if (seen >= (unsigned)(panel + 1)) break;
                if (++spins > BASQ_SPIN_LIMIT) { seen = BASQ_CHOLQR_ABORT; break; }     // never in a healthy run
                __builtin_amdgcn_s_sleep(1);
            }
            s_seen = seen;
        }
        __syncthreads();                                  // (also: Y is loaded, the previous panel's block is solved)
        if (s_seen >= BASQ_CHOLQR_ABORT) {                // uniform: failed pivot (info[0] says so) or a time-out
            if (s_seen == BASQ_CHOLQR_ABORT && tid == 0 && blockIdx.x == 1 && info[0] == 0) atomicMax(info, q + 1000);
            return;
        }
        if (FULL) {
            for (int e = tid; e < (q - j0) * NB; e += 512) {   // column panel `panel` of L -> LDS (L1-bypassing loads)
                const int i = j0 + e / NB, cc = e % NB;
                if (cc < nb && j0 + cc <= i)
                    Lsh[(size_t)i * q + j0 + cc] = __longlong_as_double((long long)__hip_atomic_load(
                        (basq_gu64*)(G + (size_t)i * q + j0 + cc), BASQ_RLX_AGENT));
            }
        } else {
            // rows j0 .. j0 + nb - 1 of L, whole (their left parts were published with the earlier panels, whose drains
            // precede this panel's in the publishing wave's program order)
            const int w = j0 + nb;
            for (int e = tid; e < nb * w; e += 512) {
                const int cc = e / w, k = e - cc * w;
                if (k <= j0 + cc)
                    Lsh[(size_t)cc * q + k] = __longlong_as_double((long long)__hip_atomic_load(
                        (basq_gu64*)(G + (size_t)(j0 + cc) * q + k), BASQ_RLX_AGENT));
            }
        }
        const int lbase = FULL ? j0 : 0;                   // LDS row of L's row j0
        __syncthreads();
        if (c < nb) {
            const double* lrow = Lsh + (size_t)(lbase + c) * q;
            const double* yrow = Y + r * ld;
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
            int k = 0;
            for (; k + 3 < j0; k += 4) {
                s0 = __builtin_fma(yrow[k], lrow[k], s0);
                s1 = __builtin_fma(yrow[k + 1], lrow[k + 1], s1);
                s2 = __builtin_fma(yrow[k + 2], lrow[k + 2], s2);
                s3 = __builtin_fma(yrow[k + 3], lrow[k + 3], s3);
            }
            for (; k < j0; ++k) s0 = __builtin_fma(yrow[k], lrow[k], s0);
            Y[r * ld + j0 + c] -= (s0 + s1) + (s2 + s3);
        }
        __syncthreads();
        if (tid < 64) {                                   // one thread per row: the 8 x 8 block by forward substitution
            double y[NB];
#pragma unroll
            for (int cc = 0; cc < NB; ++cc) y[cc] = (cc < nb) ? Y[tid * ld + j0 + cc] : 0.0;
#pragma unroll
            for (int cc = 0; cc < NB; ++cc) {
                if (cc < nb) {
                    const double* lrow = Lsh + (size_t)(lbase + cc) * q + j0;
                    double v = y[cc];
#pragma unroll
                    for (int k = 0; k < cc; ++k) v -= y[k] * lrow[k];
                    y[cc] = v / lrow[cc];
                }
            }
#pragma unroll
            for (int cc = 0; cc < NB; ++cc)
                if (cc < nb) Y[tid * ld + j0 + cc] = y[cc];
        }
    }
    __syncthreads();
    for (int e = tid; e < 64 * q; e += 512) {
        const int rr = e / q, cc = e - rr * q;
        if (rr < nr) Qo[(r0 + rr) * ldq + cc] = Y[rr * ld + cc];
    }
}

// ------------------------------------------------------------------------------------------------
// Block sums of a dense per-pair matrix handed over by the caller:
//     E[j][s] += scale * sum_{p in chunk, set(p) = s} mu_p * C[j][p]        (SQ = false)
//     E[j][s] += scale * sum_{p in chunk, set(p) = s} mu_p * C[j][p]^2      (SQ = true)
// SQ = false is the hot loop of BASQ/_rchq.py:79-99 for an OPAQUE kernel callable (the reference's `kernel` argument is
// any Python callable; C = kernel(pts_nys, chunk of candidates) is evaluated by the caller, on the device);
// SQ = true is WSABI-M's 0.5 cov^2 term (BASQ/_wsabi.py:240-242).
// C [m, nc] holds the values of the Nystrom rows against nc consecutive candidates whose first global position is pg0.
// A work-group owns JR rows and all S sets: thread = set, so consecutive lanes read consecutive candidates of a
// row (coalesced 512-B wave loads) and every (row, set) sum runs in position order; chunks are launched in position
// order, hence a fixed summation order overall.  HBM-bound by construction: 8 B per pair, read once.
// ------------------------------------------------------------------------------------------------
template <int JR, bool SQ>
__global__ void __launch_bounds__(256) dense_blocksum_kernel(const double* __restrict__ C, int m, long long nc,
                                                             long long ldc, const double* __restrict__ mu,
                                                             long long pg0, long long n_full, int S, double scale,
                                                             double* __restrict__ E, double* __restrict__ T) {
    const int j0 = blockIdx.x * JR;
    const double* rows[JR];
#pragma unroll
    for (int jr = 0; jr < JR; ++jr) rows[jr] = C + (long long)((j0 + jr < m) ? (j0 + jr) : (m - 1)) * ldc;
    const long long blk_end = (pg0 + nc < n_full) ? (pg0 + nc) : n_full;     // end of the block positions of this chunk
    const bool do_tot = T != nullptr && blockIdx.x == 0;                      // the set weights ride along in work-group 0
    for (int s = threadIdx.x; s < S; s += 256) {
        double acc[JR];
        double wacc = 0.0;
#pragma unroll
        for (int jr = 0; jr < JR; ++jr) acc[jr] = 0.0;
        long long o = ((s - pg0 % S) % S + S) % S;                            // first chunk offset whose position = s mod S
        const long long oe = blk_end - pg0;
        // two positions per trip: 2 JR + 2 independent loads in flight per lane
        for (; o + S < oe; o += 2 * (long long)S) {
            const double w0 = mu[o], w1 = mu[o + S];
            wacc += w0;
            wacc += w1;
            double c0[JR], c1[JR];
#pragma unroll
            for (int jr = 0; jr < JR; ++jr) { c0[jr] = rows[jr][o]; c1[jr] = rows[jr][o + S]; }
#pragma unroll
            for (int jr = 0; jr < JR; ++jr) {
                acc[jr] = __builtin_fma(SQ ? w0 * c0[jr] : w0, c0[jr], acc[jr]);
                acc[jr] = __builtin_fma(SQ ? w1 * c1[jr] : w1, c1[jr], acc[jr]);
            }
        }
        for (; o < oe; o += S) {
            const double w0 = mu[o];
            wacc += w0;
#pragma unroll
            for (int jr = 0; jr < JR; ++jr) {
                const double c = rows[jr][o];
                acc[jr] = __builtin_fma(SQ ? w0 * c : w0, c, acc[jr]);
            }
        }
        if (s == S - 1) {                                                     // ragged tail: positions >= n_full
            long long t = ((n_full > pg0) ? n_full : pg0) - pg0;
            for (; t < nc; ++t) {
                const double w0 = mu[t];
                wacc += w0;
#pragma unroll
                for (int jr = 0; jr < JR; ++jr) {
                    const double c = rows[jr][t];
                    acc[jr] = __builtin_fma(SQ ? w0 * c : w0, c, acc[jr]);
                }
            }
        }
#pragma unroll
        for (int jr = 0; jr < JR; ++jr)
            if (j0 + jr < m) E[(long long)(j0 + jr) * S + s] += scale * acc[jr];
        if (do_tot) T[s] += wacc;
    }
}

// The same sums with 16 bytes per lane (round 4): a thread owns a PAIR of neighbouring sets (2t, 2t+1) -- two consecutive
// candidates of a row = one 16-byte load -- in one of NS position slices (slice k takes the blocks k, k + NS, ... of the
// chunk), JR = 8 rows per work-group and two blocks per trip: 16 independent 16-byte loads in flight per lane (the 8-byte
// form had ~10 of 8 bytes and ran at 3.0 TB/s, latency-bound: profiles/r03_k_cfg4_opaque_kernel_stats.csv).  The slices
// are added in slice order through LDS, the ragged tail (all of it belongs to set S-1) is spread over the work-group
// and added wave by wave: every sum has a fixed order.  Needs S even and an even first position (the engine cuts its
// chunks that way); everything else takes the kernel above.
struct __attribute__((aligned(8))) DPair { double x, y; };
// The kernel values are read exactly once: non-temporal 16-byte loads (no L2 allocation; the default whenever the pairs are
// 16-byte aligned) -- the block sums E, which every chunk reads and writes back, then keep the cache.
typedef double DVec2 __attribute__((ext_vector_type(2)));
template <bool NT>
__device__ __forceinline__ DPair load_pair(const double* p) {
    if (NT) {
        const DVec2 v = __builtin_nontemporal_load(reinterpret_cast<const DVec2*>(p));
        return DPair{v.x, v.y};
    }
    return *reinterpret_cast<const DPair*>(p);
}

template <int JR, bool SQ, bool NT>
__global__ void __launch_bounds__(1024) dense_blocksum_pairs_kernel(const double* __restrict__ C, int m, long long nc,
                                                                    long long ldc, const double* __restrict__ mu,
                                                                    long long pg0, long long n_full, int S, int NS,
                                                                    double scale, double* __restrict__ E,
                                                                    double* __restrict__ T) {
    extern __shared__ double dbs_red[];                          // [NS][JR][S] slice partials, then [waves][JR] tail partials
    const int half = S >> 1;
    const int t = threadIdx.x % half, k = threadIdx.x / half;   // set pair, position slice (blockDim.x = half * NS)
    const int j0 = blockIdx.x * JR;
    const double* rows[JR];
#pragma unroll
    for (int jr = 0; jr < JR; ++jr) rows[jr] = C + (long long)((j0 + jr < m) ? (j0 + jr) : (m - 1)) * ldc;
    const long long blk_end = (pg0 + nc < n_full) ? (pg0 + nc) : n_full;
    const long long oe = blk_end - pg0;                          // block positions of this chunk: offsets [0, oe)
    const int ph = (int)(pg0 % S);                               // even
    long long o = (long long)(((2 * t - ph) % S + S) % S) + (long long)k * S;   // first offset of set 2t in slice k
    const long long step = (long long)NS * S;
    double a0[JR], a1[JR];
    double wa0 = 0.0, wa1 = 0.0;                                 // set weights (consumed by work-group 0 only)
#pragma unroll
    for (int jr = 0; jr < JR; ++jr) { a0[jr] = 0.0; a1[jr] = 0.0; }
    for (; o + step + 1 < oe; o += 2 * step) {                   // two blocks per trip: 2 JR + 2 loads of 16 bytes in flight
        const DPair w0 = *reinterpret_cast<const DPair*>(mu + o), w1 = *reinterpret_cast<const DPair*>(mu + o + step);
        wa0 += w0.x; wa1 += w0.y;
        wa0 += w1.x; wa1 += w1.y;
        DPair c0[JR], c1[JR];
#pragma unroll
        for (int jr = 0; jr < JR; ++jr) {
            c0[jr] = load_pair<NT>(rows[jr] + o);
            c1[jr] = load_pair<NT>(rows[jr] + o + step);
        }
#pragma unroll
        for (int jr = 0; jr < JR; ++jr) {
            a0[jr] = __builtin_fma(SQ ? w0.x * c0[jr].x : w0.x, c0[jr].x, a0[jr]);
            a1[jr] = __builtin_fma(SQ ? w0.y * c0[jr].y : w0.y, c0[jr].y, a1[jr]);
            a0[jr] = __builtin_fma(SQ ? w1.x * c1[jr].x : w1.x, c1[jr].x, a0[jr]);
            a1[jr] = __builtin_fma(SQ ? w1.y * c1[jr].y : w1.y, c1[jr].y, a1[jr]);
        }
    }
    for (; o + 1 < oe; o += step) {
        const DPair w0 = *reinterpret_cast<const DPair*>(mu + o);
        wa0 += w0.x; wa1 += w0.y;
#pragma unroll
        for (int jr = 0; jr < JR; ++jr) {
            const DPair c = load_pair<NT>(rows[jr] + o);
            a0[jr] = __builtin_fma(SQ ? w0.x * c.x : w0.x, c.x, a0[jr]);
            a1[jr] = __builtin_fma(SQ ? w0.y * c.y : w0.y, c.y, a1[jr]);
        }
    }
    if (o < oe) {                                                // the chunk's block positions end inside this pair
        const double w0 = mu[o];
        wa0 += w0;
#pragma unroll
        for (int jr = 0; jr < JR; ++jr) {
            const double c = rows[jr][o];
            a0[jr] = __builtin_fma(SQ ? w0 * c : w0, c, a0[jr]);
        }
    }
    // slices -> slice 0, in slice order (row JR of a slice's LDS image = its set weights)
    const bool do_tot = T != nullptr && blockIdx.x == 0;
    if (NS > 1) {
        if (k > 0) {
#pragma unroll
            for (int jr = 0; jr < JR; ++jr) {
                double* dst = dbs_red + ((long long)(k * (JR + 1) + jr)) * S + 2 * t;
                dst[0] = a0[jr];
                dst[1] = a1[jr];
            }
            double* dw = dbs_red + ((long long)(k * (JR + 1) + JR)) * S + 2 * t;
            dw[0] = wa0;
            dw[1] = wa1;
        }
        __syncthreads();
        if (k == 0) {
            for (int kk = 1; kk < NS; ++kk) {
#pragma unroll
                for (int jr = 0; jr < JR; ++jr) {
                    const double* src = dbs_red + ((long long)(kk * (JR + 1) + jr)) * S + 2 * t;
                    a0[jr] += src[0];
                    a1[jr] += src[1];
                }
                const double* sw = dbs_red + ((long long)(kk * (JR + 1) + JR)) * S + 2 * t;
                wa0 += sw[0];
                wa1 += sw[1];
            }
        }
    }
    // ragged tail: offsets [tl, nc) all belong to set S-1 (BASQ/_rchq.py:91-99)
    const long long tl = ((n_full > pg0) ? n_full : pg0) - pg0;
    if (tl < nc) {                                               // work-group uniform
        double tt[JR];
        double tw = 0.0;
#pragma unroll
        for (int jr = 0; jr < JR; ++jr) tt[jr] = 0.0;
        for (long long q = tl + threadIdx.x; q < nc; q += blockDim.x) {
            const double w0 = mu[q];
            tw += w0;
#pragma unroll
            for (int jr = 0; jr < JR; ++jr) {
                const double c = rows[jr][q];
                tt[jr] = __builtin_fma(SQ ? w0 * c : w0, c, tt[jr]);
            }
        }
#pragma unroll
        for (int jr = 0; jr < JR; ++jr)
            for (int sh = 32; sh >= 1; sh >>= 1) tt[jr] += __shfl_xor(tt[jr], sh, 64);
        for (int sh = 32; sh >= 1; sh >>= 1) tw += __shfl_xor(tw, sh, 64);
        __syncthreads();                                         // the slice partials have been consumed
        const int wave = threadIdx.x >> 6, nwaves = (blockDim.x + 63) >> 6;
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int jr = 0; jr < JR; ++jr) dbs_red[wave * (JR + 1) + jr] = tt[jr];
            dbs_red[wave * (JR + 1) + JR] = tw;
        }
        __syncthreads();
        if (k == 0 && t == half - 1)
            for (int w = 0; w < nwaves; ++w) {
#pragma unroll
                for (int jr = 0; jr < JR; ++jr) a1[jr] += dbs_red[w * (JR + 1) + jr];
                wa1 += dbs_red[w * (JR + 1) + JR];
            }
    }
    if (k == 0) {
#pragma unroll
        for (int jr = 0; jr < JR; ++jr)
            if (j0 + jr < m) {
                double* e = E + (long long)(j0 + jr) * S + 2 * t;
                e[0] += scale * a0[jr];
                e[1] += scale * a1[jr];
            }
        if (do_tot) {
            T[2 * t] += wa0;
            T[2 * t + 1] += wa1;
        }
    }
}

// The likelihood noise inside WSABI-M's squared covariance, per candidate (BASQ/_gp.py:275-276 under _wsabi.py:240-242):
// predictive_covariance adds the noise to entry [kappa][kappa] of every kernel block -- candidate p meets it on the Nystrom
// row kappa = its position inside its block (p % S below n_full, p - n_full in the ragged remainder) -- so
//     0.5 (c + noise)^2 = 0.5 c^2 + (noise c + 0.5 noise^2)      on that one row,     c = cov(nys_kappa, x_p) without noise.
// The first term is a plain per-pair block sum (basq_blocksum_sq_f64 with noise = 0: it regroups over the rounds of an epoch
// like every other block sum); this kernel evaluates the bracket, one thread per candidate (0 where kappa >= m):
//     out[p] = noise * (outputscale k(nys_kappa, x_p) - sum_o bmatT[o][kappa] kobs[o][p]) + 0.5 noise^2
template <int FAM>
__global__ void cov_diag_kernel(const double* __restrict__ nys, int kp, int m, const double* __restrict__ cand,
                                long long Rl, long long off, long long n_full, int S, const double* __restrict__ bmatT,
                                long long ldb, const double* __restrict__ kobs, long long ldk, int n_obs,
                                double outputscale, double noise, double* __restrict__ out) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= Rl) return;
    const long long pg = off + p;
    const long long kappa = (pg < n_full) ? (pg % S) : (pg - n_full);
    if (kappa >= m) {
        out[p] = 0.0;
        return;
    }
    const double* a = nys + kappa * kp;
    const double* b = cand + p * kp;
    double D = 0.0;
    for (int k = 0; k < kp; ++k) D = __builtin_fma(a[k], b[k], D);
    double corr = 0.0;
    for (int o = 0; o < n_obs; ++o) corr = __builtin_fma(bmatT[(long long)o * ldb + kappa], kobs[(long long)o * ldk + p], corr);
    const double c = __builtin_fma(outputscale, kernel_from_arg<FAM>(D), -corr);
    out[p] = __builtin_fma(noise, c, 0.5 * noise * noise);
}

// ------------------------------------------------------------------------------------------------
// Box-Muller transform of torch's CPU ``normal_fill`` (aten/native/cpu/DistributionTemplates.h): blocks of 16
// uniforms -> 16 normals (u1 = 1 - u[j], u2 = u[j+8]; r = sqrt(-2 log u1), t = 2 pi u2; out[j] = r cos t,
// out[j+8] = r sin t).  The Gaussian test matrix of torch.svd_lowrank is one ``torch.randn`` from the CPU
// generator; drawing the SAME uniforms with ``torch.rand`` (identical generator consumption, verified by
// tests/test_host_logic.py) and transforming them here takes the ~12 ms of scalar libm calls off the host.
// When n % 16 != 0 torch recomputes the last 16 values from 16 fresh uniforms (u_tail).
// ------------------------------------------------------------------------------------------------
__global__ void box_muller_kernel(const double* __restrict__ u, long long nblk, double* __restrict__ out) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;     // one (block of 16, j < 8) pair per thread
    if (t >= nblk * 8) return;
    const long long blk = t >> 3;
    const int j = (int)(t & 7);
    const double* src = u + blk * 16;
    double* dst = out + blk * 16;
    const double u1 = 1.0 - src[j];
    const double u2 = src[j + 8];
    const double radius = sqrt(-2.0 * log(u1));
    const double theta = 2.0 * 3.14159265358979323846 * u2;
    dst[j] = radius * cos(theta);
    dst[j + 8] = radius * sin(theta);
}

__global__ void axpb_strided_kernel(const double* __restrict__ x, long long n, long long stride, double a, double b,
                                    double* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = __builtin_fma(a, x[i * stride], b);
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
static inline bool spec_ok(const basq_kernel_spec* s) {
    return s && s->d >= 1 && s->d <= BASQ_MAX_DIM && s->lengthscale > 0.0 && s->family >= 0 && s->family <= 2;
}
// exponential scheme of the fused block sums: the accurate one (1e-17) where the caller asks for it (GP posteriors), else
// the build's default (2: 2.5e-14, one fp64 instruction less per kernel value)
static inline int blocksum_exp_scheme(const basq_kernel_spec* s) {
    return (s->flags & BASQ_SPEC_ACCURATE_EXP) ? 1 : BASQ_BLOCKSUM_EXP_SCHEME;
}

extern "C" {

const char* basq_strerror(int code) {
    switch (code) {
        case BASQ_OK: return "ok";
        case BASQ_EINVAL: return "invalid argument";
        case BASQ_ELAUNCH: return "HIP kernel launch failed";
        case BASQ_EUNSUPPORTED: return "kernel family or size not supported";
    }
    return "unknown error";
}

int basq_abi_version(void) { return BASQ_ABI_VERSION; }

int basq_kp(int d) {
    if (d < 1 || d > BASQ_MAX_DIM) return BASQ_EINVAL;
    return ((d + 2 + 3) / 4) * 4;
}

// Measurement aid (bench.py's roofline): the shader clock over the next n * period_us microseconds, one wave counting
// s_memtime cycles per period of the constant 100-MHz s_memrealtime.  Launched on a SECOND stream beside the kernel of
// interest (one 64-thread work-group fits next to anything), it sees the clock that kernel runs at: under a full fp64 load
// that follows >= 5 ms without one (a single-work-group reduction chain counts as without) the power manager holds the chip
// at ~2.05 GHz and raises it by only ~20 MHz per ms (tools/clock_probe.hip) -- an otherwise idle chip reads 2.43 GHz.
__global__ void shader_clock_kernel(double* out, int n, unsigned long long period_ticks) {
    unsigned long long r0 = wall_clock64(), c0 = clock64();
    for (int k = 0; k < n; ++k) {
        unsigned long long r1 = r0;
        while (r1 - r0 < period_ticks) {
            __builtin_amdgcn_s_sleep(8);
            r1 = wall_clock64();
        }
        const unsigned long long c1 = clock64();
        if (threadIdx.x == 0) out[k] = 100.0 * (double)(c1 - c0) / (double)(r1 - r0);
        r0 = r1;
        c0 = c1;
    }
}

int basq_shader_clock_mhz(double* out, int n, int period_us, void* stream) {
    if (!out || n < 1 || period_us < 1 || (long long)n * period_us > 1000000) return BASQ_EINVAL;   // at most 1 s of sampling
    hipLaunchKernelGGL(shader_clock_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out, n, 100ull * period_us);
    BASQ_CHECK_LAUNCH();
    return BASQ_OK;
}

int basq_col_mean_f64(const double* X, int64_t n, int d, double* mean, void* stream) {
    if (!X || !mean || n < 0 || d < 1 || d > BASQ_MAX_DIM) return BASQ_EINVAL;
    hipLaunchKernelGGL(col_mean_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, X, (long long)n, d, mean);
    BASQ_CHECK_LAUNCH();
    return BASQ_OK;
}

int basq_pack_points_f64(const basq_kernel_spec* spec, const double* X, int64_t n, const double* center, int role,
                         double* out, void* stream) {
    if (!spec_ok(spec) || !out || n < 0 || (n > 0 && !X) || (role != BASQ_ROLE_A && role != BASQ_ROLE_B))
        return BASQ_EINVAL;
    if (n == 0) return BASQ_OK;
    const int kp = basq_kp(spec->d);
    const int ppb = (kp <= 28) ? 256 : 128;                 // the LDS tile stays under 64 KB
    hipLaunchKernelGGL(pack_points_kernel, dim3((unsigned)((n + ppb - 1) / ppb)), dim3(256),
                       (size_t)ppb * (kp | 1) * sizeof(double), (hipStream_t)stream, X, (long long)n, spec->d, kp, center,
                       1.0 / spec->lengthscale, role, out, ppb);
    BASQ_CHECK_LAUNCH();
    return BASQ_OK;
}

int basq_gram_f64(const basq_kernel_spec* spec, const double* packA, int64_t na, const double* packB, int64_t nb,
                  double* K, int64_t ldk, void* stream) {
    if (!spec_ok(spec) || na < 0 || nb < 0 || ldk < nb) return BASQ_EINVAL;
    if (na == 0 || nb == 0) return BASQ_OK;
    if (!packA || !packB || !K) return BASQ_EINVAL;
    const int kk = basq_kp(spec->d) / 4;
    hipStream_t st = (hipStream_t)stream;
    const double sc = spec->outputscale;
    switch (kk) {
        case 1: return dispatch_gram_fam<1>(spec->family, packA, na, packB, nb, sc, K, ldk, st);
        case 2: return dispatch_gram_fam<2>(spec->family, packA, na, packB, nb, sc, K, ldk, st);
        case 3: return dispatch_gram_fam<3>(spec->family, packA, na, packB, nb, sc, K, ldk, st);
        case 4: return dispatch_gram_fam<4>(spec->family, packA, na, packB, nb, sc, K, ldk, st);
        case 5: return dispatch_gram_fam<5>(spec->family, packA, na, packB, nb, sc, K, ldk, st);
        case 6: return dispatch_gram_fam<6>(spec->family, packA, na, packB, nb, sc, K, ldk, st);
        case 7: return dispatch_gram_fam<7>(spec->family, packA, na, packB, nb, sc, K, ldk, st);
        case 8: return dispatch_gram_fam<8>(spec->family, packA, na, packB, nb, sc, K, ldk, st);
        case 9: return dispatch_gram_fam<9>(spec->family, packA, na, packB, nb, sc, K, ldk, st);
        case 10: return dispatch_gram_fam<10>(spec->family, packA, na, packB, nb, sc, K, ldk, st);
    }
    return BASQ_EUNSUPPORTED;
}

static int blocksum_impl(const basq_kernel_spec* spec, const double* nys, int32_t m, const double* cand,
                         const double* mu, const double* wx, int64_t Rl, int64_t off, int64_t n_full, int32_t S,
                         int32_t n_chunks, int32_t class_mod, int32_t class0, double* Xpart, double* totpart,
                         void* stream, int xs) {
    if (!spec_ok(spec) || !nys || !cand || !mu || !Xpart) return BASQ_EINVAL;
    if (m < 1 || Rl < 1 || off < 0 || n_full < 0 || S < 1 || n_chunks < 1) return BASQ_EINVAL;
    if (n_full % S != 0) return BASQ_EINVAL;
    if (class_mod < 0 || class0 < 0 || (class_mod > 0 && class0 + n_chunks > class_mod)) return BASQ_EINVAL;
    if (class_mod > 0 && off + Rl > n_full) return BASQ_EINVAL;   // residue classes cover full blocks only (no ragged tail)
    BlocksumArgs A;
    A.nys = nys; A.cand = cand; A.mu = mu; A.wx = wx; A.Xpart = Xpart; A.totpart = totpart;
    A.Rl = Rl; A.off = off; A.n_full = n_full; A.m = m; A.S = S; A.n_chunks = n_chunks;
    A.class_mod = class_mod; A.class0 = class0;
    A.geo = nullptr; A.geo_mode = 0;
    A.n_stiles = (S + 15) / 16;
    // global blocks that intersect [off, min(off+Rl, n_full))
    const long long lim = (off + Rl < n_full) ? (off + Rl) : n_full;
    if (lim > off) {
        A.blk_lo = off / S;
        A.blk_hi = (lim + S - 1) / S;
    } else {
        A.blk_lo = 0;
        A.blk_hi = 0;
    }
    const long long nblk = A.blk_hi - A.blk_lo;
    A.blk_per_chunk = (nblk + n_chunks - 1) / n_chunks;
    if (A.blk_per_chunk < 1) A.blk_per_chunk = 1;
    return dispatch_blocksum(basq_kp(spec->d) / 4, spec->family, A, (hipStream_t)stream, xs);
}

int basq_blocksum_f64(const basq_kernel_spec* spec, const double* nys, int32_t m, const double* cand,
                      const double* mu, const double* wx, int64_t Rl, int64_t off, int64_t n_full, int32_t S,
                      int32_t n_chunks, int32_t class_mod, int32_t class0, double* Xpart, double* totpart, void* stream) {
    if (!totpart) return BASQ_EINVAL;
    return blocksum_impl(spec, nys, m, cand, mu, wx, Rl, off, n_full, S, n_chunks, class_mod, class0, Xpart, totpart,
                         stream, blocksum_exp_scheme(spec));
}

int basq_cov_diag_f64(const basq_kernel_spec* spec, const double* nys, int32_t m, const double* cand, int64_t Rl,
                      int64_t off, int64_t n_full, int32_t S, const double* bmatT, int64_t ldb, const double* kobs,
                      int64_t ldk, int32_t n_obs, double noise, double* out, void* stream) {
    if (!spec_ok(spec) || !nys || !cand || !bmatT || !kobs || !out) return BASQ_EINVAL;
    if (m < 1 || Rl < 0 || off < 0 || n_full < 0 || S < 1 || n_obs < 1 || n_full % S != 0 || ldb < m || ldk < Rl)
        return BASQ_EINVAL;
    if (Rl == 0) return BASQ_OK;
    const int kp = basq_kp(spec->d);
    const dim3 grid((unsigned)((Rl + 255) / 256)), block(256);
#define BASQ_COV_DIAG(FAM)                                                                                              \
    hipLaunchKernelGGL((cov_diag_kernel<FAM>), grid, block, 0, (hipStream_t)stream, nys, kp, m, cand, (long long)Rl,    \
                       (long long)off, (long long)n_full, S, bmatT, (long long)ldb, kobs, (long long)ldk, n_obs,         \
                       spec->outputscale, noise, out)
    switch (spec->family) {
        case BASQ_FAMILY_RBF: BASQ_COV_DIAG(BASQ_FAMILY_RBF); break;
        case BASQ_FAMILY_MATERN52: BASQ_COV_DIAG(BASQ_FAMILY_MATERN52); break;
        case BASQ_FAMILY_MATERN32: BASQ_COV_DIAG(BASQ_FAMILY_MATERN32); break;
        default: return BASQ_EUNSUPPORTED;
    }
#undef BASQ_COV_DIAG
    BASQ_CHECK_LAUNCH();
    return BASQ_OK;
}

int basq_blocksum_sq_f64(const basq_kernel_spec* spec, const double* nys, int32_t m, const double* cand,
                         const double* mu, int64_t Rl, int64_t off, int64_t n_full, int32_t S, int32_t n_chunks,
                         int32_t class_mod, int32_t class0,
                         const double* bmatT, int64_t ldb, const double* kobs, int64_t ldk, int32_t n_obs, double noise,
                         double* Epart, void* stream) {
    if (!spec_ok(spec) || !nys || !cand || !mu || !bmatT || !kobs || !Epart) return BASQ_EINVAL;
    if (m < 1 || Rl < 1 || off < 0 || n_full < 0 || S < 1 || n_chunks < 1 || n_obs < 1) return BASQ_EINVAL;
    if (n_full % S != 0) return BASQ_EINVAL;
    if (class_mod < 0 || class0 < 0 || (class_mod > 0 && class0 + n_chunks > class_mod)) return BASQ_EINVAL;
    if (class_mod > 0 && off + Rl > n_full) return BASQ_EINVAL;   // residue classes cover full blocks only
    const int jt = BASQ_JT_FOR(basq_kp(spec->d) / 4);
    if (ldb < (((int64_t)m + 16 * jt - 1) / (16 * jt)) * (16 * jt) || ldk < Rl) return BASQ_EINVAL;   // fragment reads stay inside
    BlocksumArgs A;
    A.nys = nys; A.cand = cand; A.mu = mu; A.wx = nullptr; A.Xpart = Epart; A.totpart = nullptr;
    A.Rl = Rl; A.off = off; A.n_full = n_full; A.m = m; A.S = S; A.n_chunks = n_chunks;
    A.class_mod = class_mod; A.class0 = class0;
    A.geo = nullptr; A.geo_mode = 0;
    A.n_stiles = (S + 15) / 16;
    const long long lim = (off + Rl < n_full) ? (off + Rl) : n_full;
    if (lim > off) {
        A.blk_lo = off / S;
        A.blk_hi = (lim + S - 1) / S;
    } else {
        A.blk_lo = 0;
        A.blk_hi = 0;
    }
    const long long nblk = A.blk_hi - A.blk_lo;
    A.blk_per_chunk = (nblk + n_chunks - 1) / n_chunks;
    if (A.blk_per_chunk < 1) A.blk_per_chunk = 1;
    SqArgs Q;
    Q.bmatT = bmatT; Q.kobs = kobs; Q.ldb = ldb; Q.ldk = ldk; Q.ko = (n_obs + 3) / 4;
    Q.outputscale = spec->outputscale; Q.noise = noise;
    return dispatch_blocksum_sq(basq_kp(spec->d) / 4, spec->family, A, Q, (hipStream_t)stream);
}

// Next round's class partials from this round's, without touching a candidate (see include/basq_hip.h):
//   Tout[c'][j][par * H + k] = (Tin[2 c' + par][j][kept[k]] * w_star[k]) / tot[kept[k]],   H = n_keep = S / 2
__global__ void regroup_classes_kernel(const double* __restrict__ Tin, int rows, int S, int C, const int* __restrict__ kept,
                                       const double* __restrict__ w_star, const double* __restrict__ tot,
                                       double* __restrict__ Tout) {
#pragma clang fp contract(off)
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long n = (long long)(C / 2) * rows * S;
    if (e >= n) return;
    const int sp = (int)(e % S);
    const long long rest = e / S;
    const int j = (int)(rest % rows), cp = (int)(rest / rows);
    const int H = S / 2, par = sp / H, k = sp - par * H;
    const int s = kept[k];
    if ((unsigned)s >= (unsigned)S) {          // launched before the host checked n_keep: entries past it are stale
        Tout[e] = 0.0;
        return;
    }
    const double v = Tin[((long long)(2 * cp + par) * rows + j) * S + s];
    Tout[e] = (v * w_star[k]) / tot[s];                                        // the order of BASQ/_rchq.py:113-114
}

// The two launches that follow every elimination inside an epoch, as ONE (round 4): the regrouping of the class messages and,
// in one extra work-group, the next round's descriptor (both read the elimination's outcome; neither reads the other's).
__global__ void regroup_round_next_kernel(const double* __restrict__ Tin, int rows, int S, int C, const int* __restrict__ kept,
                                          const double* __restrict__ w_star, const double* __restrict__ tot,
                                          double* __restrict__ Tout, int n_regroup_blocks, const long long* __restrict__ gp,
                                          const int* __restrict__ info, const int* __restrict__ keep_rank, int class_mode,
                                          int expect_half, long long* __restrict__ gn) {
#pragma clang fp contract(off)
    if ((int)blockIdx.x >= n_regroup_blocks) {
        if (threadIdx.x < 64) round_next_body(threadIdx.x, gp, info, keep_rank, S, class_mode, expect_half, gn);
        return;
    }
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long n = (long long)(C / 2) * rows * S;
    if (e >= n) return;
    const int sp = (int)(e % S);
    const long long rest = e / S;
    const int j = (int)(rest % rows), cp = (int)(rest / rows);
    const int H = S / 2, par = sp / H, k = sp - par * H;
    const int s = kept[k];
    if ((unsigned)s >= (unsigned)S) {
        Tout[e] = 0.0;
        return;
    }
    const double v = Tin[((long long)(2 * cp + par) * rows + j) * S + s];
    Tout[e] = (v * w_star[k]) / tot[s];                                        // the order of BASQ/_rchq.py:113-114
}

int basq_regroup_round_next_f64(const double* Tin, int32_t rows, int32_t S, int32_t C, const int32_t* kept,
                                const double* w_star, const double* tot, double* Tout, const int64_t* geo,
                                const int32_t* info, const int32_t* keep_rank, int32_t class_mode, int32_t expect_half,
                                int64_t* geo_next, void* stream) {
    if (!Tin || !kept || !w_star || !tot || !Tout || rows < 1 || S < 2 || (S & 1) || C < 2 || (C & 1)) return BASQ_EINVAL;
    if (!geo || !info || !keep_rank || !geo_next) return BASQ_EINVAL;
    const long long n = (long long)(C / 2) * rows * S;
    const long long nb = (n + 255) / 256;
    if (nb + 1 > 0x7fffffffLL) return BASQ_EINVAL;
    hipLaunchKernelGGL(regroup_round_next_kernel, dim3((unsigned)(nb + 1)), dim3(256), 0, (hipStream_t)stream, Tin, rows, S, C,
                       kept, w_star, tot, Tout, (int)nb, (const long long*)geo, info, keep_rank, class_mode, expect_half,
                       (long long*)geo_next);
    BASQ_CHECK_LAUNCH();
    return BASQ_OK;
}

int basq_regroup_classes_f64(const double* Tin, int32_t rows, int32_t S, int32_t C, const int32_t* kept,
                             const double* w_star, const double* tot, double* Tout, void* stream) {
    if (!Tin || !kept || !w_star || !tot || !Tout || rows < 1 || S < 2 || (S & 1) || C < 2 || (C & 1)) return BASQ_EINVAL;
    const long long n = (long long)(C / 2) * rows * S;
    hipLaunchKernelGGL(regroup_classes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, Tin,
                       rows, S, C, kept, w_star, tot, Tout);
    BASQ_CHECK_LAUNCH();
    return BASQ_OK;
}

int basq_kernel_matvec_f64(const basq_kernel_spec* spec, const double* packA, int64_t na, const double* packB,
                           int64_t nb, const double* v, double bias, double* out, void* stream) {
    // blocksum with a single set and everything in the "tail": out[i] = sum_j k(A_i, B_j) v_j.
    if (!spec_ok(spec) || !packA || !packB || !v || !out || na < 1 || nb < 1 || na > 0x7fffffffLL)
        return BASQ_EINVAL;
    int rc = blocksum_impl(spec, packA, (int32_t)na, packB, v, nullptr, nb, 0, 0, 1, 1, 0, 0, out, nullptr, stream, 1);
    if (rc != BASQ_OK) return rc;
    hipLaunchKernelGGL(axpb_strided_kernel, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out,
                       (long long)na, 1LL, spec->outputscale, bias, out);
    BASQ_CHECK_LAUNCH();
    return BASQ_OK;
}

int basq_project_f64(const double* Ut, int32_t q, int32_t m, const double* Xpart, const double* totpart,
                     int32_t n_chunks, int32_t S, double outputscale, int32_t ksplit, double* work, double* out,
                     void* stream) {
    if (!Ut || !Xpart || !totpart || !work || !out || q < 1 || m < 1 || S < 1 || n_chunks < 1 || ksplit < 1)
        return BASQ_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    // work = [ Xsum (m*S, only if n_chunks > 1) | ksplit slabs of q*S ]
    const long long nX = (long long)m * S;
    const double* X = Xpart;
    double* slabs = work;
    if (n_chunks > 1) {
        if ((nX & 1) || (((uintptr_t)Xpart | (uintptr_t)work) & 15)) return BASQ_EINVAL;   // double2 path
        hipLaunchKernelGGL(chunk_sum_kernel, dim3((unsigned)((nX / 2 + 255) / 256)), dim3(256), 0, st, Xpart, nX, n_chunks,
                           work);
        BASQ_CHECK_LAUNCH();
        X = work;
        slabs = work + nX;
    }
    int kslice = (m + ksplit - 1) / ksplit;
    kslice = ((kslice + 15) / 16) * 16;
    const int nz = (m + kslice - 1) / kslice;          // <= ksplit slabs
    // 7 row tiles per wave (112 rows: all of U at n = 100): one B fragment feeds 7 MFMAs instead of 1
    dim3 grid((unsigned)((q + 111) / 112), (unsigned)((S + 63) / 64), (unsigned)nz);
    hipLaunchKernelGGL((gemm_kernel<7>), grid, dim3(256), 0, st, Ut, 1LL, (long long)q, X, (long long)S, 0LL, 1, slabs,
                       (long long)S, (long long)q * S, q, S, m, kslice, outputscale, nz, 0LL);
    BASQ_CHECK_LAUNCH();
    const int tot = (q + 1) * S;
    hipLaunchKernelGGL(project_reduce_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, slabs, nz, q, S,
                       totpart, n_chunks, out);
    BASQ_CHECK_LAUNCH();
    return BASQ_OK;
}

// out[c][0][s] = totpart[c][s];  out[c][1+r][s] = sum_z work[c * nz + z][r][s]   (fixed order)
__global__ void project_chunks_reduce_kernel(const double* __restrict__ work, int nz, int q, int S,
                                             const double* __restrict__ totpart, int n_chunks, double* __restrict__ out) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long per = (long long)(q + 1) * S;
    if (idx >= per * n_chunks) return;
    const int c = (int)(idx / per);
    const int e = (int)(idx - (long long)c * per);
    const int r = e / S, s2 = e - r * S;
    double v = 0.0;
    if (r == 0) {
        v = totpart[(long long)c * S + s2];
    } else {
        v = ordered_strided_sum(work + ((long long)c * nz * q + (r - 1)) * S + s2, nz, (long long)q * S);
    }
    out[idx] = v;
}

// The same for slabs that come TRANSPOSED out of the tall-skinny kernel (work[(c nz + z)][s][i], i = basis row): threads run
// along i, so the slab reads are coalesced; the strided writes are 2.7 MB per epoch start.
__global__ void project_chunks_reduce_t_kernel(const double* __restrict__ work, int nz, int q, int S,
                                               const double* __restrict__ totpart, int n_chunks, double outputscale,
                                               double* __restrict__ out) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long per = (long long)(q + 1) * S;
    if (idx >= per * n_chunks) return;
    const int c = (int)(idx / per);
    const int e = (int)(idx - (long long)c * per);
    const int s2 = e / (q + 1), r = e - s2 * (q + 1);
    double v = 0.0;
    if (r == 0) {
        v = totpart[(long long)c * S + s2];
    } else {
        v = outputscale * ordered_strided_sum(work + ((long long)c * nz * S + s2) * q + (r - 1), nz, (long long)S * q);
    }
    out[((long long)c * (q + 1) + r) * S + s2] = v;
}

__global__ void sum_parts_kernel(const double* __restrict__ parts, int n_parts, long long n, double* __restrict__ out) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    out[e] = ordered_strided_sum(parts + e, n_parts, n);
}

int basq_project_chunks_f64(const double* Ut, int32_t q, int32_t m, const double* Xpart, const double* totpart,
                            int32_t n_chunks, int32_t S, double outputscale, int32_t ksplit, double* work, double* out,
                            void* stream) {
    if (!Ut || !Xpart || !totpart || !work || !out || q < 1 || m < 1 || S < 1 || n_chunks < 1 || ksplit < 1)
        return BASQ_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (q <= 208 && n_chunks <= 65535) {
        // out[c][1 + i][s] = outputscale * sum_k Ut[k][i] Xpart[c][k][s] as the batch of tall-skinny products
        // Xpart[c]^T [S, m] @ Ut [m, q]: every chunk partial (16 MB at the headline size) streams from HBM exactly once, the
        // basis slice of a K step goes through LDS once per work-group, and the products run on the full-rate 4x4x4
        // instruction (the 16 x 16-tile kernel below re-read the basis once per wave: 2 GB through L2 per epoch start).
        const int rows_per_wg = (q > 112) ? 64 : 128;
        const int rowblocks = (S + rows_per_wg - 1) / rows_per_wg;
        // K slices: as many as fit ONE round of the chip's 2048 wave slots (two 252-register waves per SIMD) -- one wave
        // too many and the launch takes two rounds (measured: 544 work-groups 249 us, 510 work-groups 183 us)
        int want = 2048 / (rowblocks * n_chunks * 4);
        if (want > m / 64) want = m / 64;                        // at least four 16-k trips per slice
        if (want > ksplit) want = ksplit;                        // (the caller sized `work` for ksplit slabs per chunk)
        if (want < 1) want = 1;
        int kslice = (m + want - 1) / want;
        kslice = ((kslice + 15) / 16) * 16;
        const int nz = (m + kslice - 1) / kslice;
        dispatch_skinny(true, S, nz, n_chunks, st, Xpart, (long long)S, (long long)m * S, Ut, (long long)q, work,
                        (long long)S * q, q, m, kslice);
        BASQ_CHECK_LAUNCH();
        const long long tot = (long long)(q + 1) * S * n_chunks;
        hipLaunchKernelGGL(project_chunks_reduce_t_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, work, nz, q,
                           S, totpart, n_chunks, outputscale, out);
        BASQ_CHECK_LAUNCH();
        return BASQ_OK;
    }
    int kslice = (m + ksplit - 1) / ksplit;
    kslice = ((kslice + 15) / 16) * 16;
    const int nz = (m + kslice - 1) / kslice;          // <= ksplit slabs per chunk
    if ((long long)n_chunks * nz > 65535) return BASQ_EINVAL;
    dim3 grid((unsigned)((q + 111) / 112), (unsigned)((S + 63) / 64), (unsigned)(n_chunks * nz));
    hipLaunchKernelGGL((gemm_kernel<7>), grid, dim3(256), 0, st, Ut, 1LL, (long long)q, Xpart, (long long)S, 0LL, 1, work,
                       (long long)S, (long long)q * S, q, S, m, kslice, outputscale, nz, (long long)m * S);
    BASQ_CHECK_LAUNCH();
    const long long tot = (long long)(q + 1) * S * n_chunks;
    hipLaunchKernelGGL(project_chunks_reduce_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, work, nz, q, S,
                       totpart, n_chunks, out);
    BASQ_CHECK_LAUNCH();
    return BASQ_OK;
}

int basq_sum_parts_f64(const double* parts, int32_t n_parts, int64_t n, double* out, void* stream) {
    if (!parts || !out || n_parts < 1 || n < 1) return BASQ_EINVAL;
    hipLaunchKernelGGL(sum_parts_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, parts, n_parts,
                       (long long)n, out);
    BASQ_CHECK_LAUNCH();
    return BASQ_OK;
}

int basq_finalize_geo_f64(const double* parts, int32_t n_parts, int32_t msg_rows, int32_t q, int32_t S,
                          const double* diagU, int64_t ld_diag, int32_t n_diag, double diag_noise, int32_t diag_wrow,
                          int32_t diag_tail_row, int32_t n_tail_diag, const int64_t* geo, double* XcarT, double* tot_out,
                          void* stream) {
    if (!parts || !XcarT || !tot_out || n_parts < 1 || q < 1 || S < 1 || msg_rows < q + 1) return BASQ_EINVAL;
    if (diag_wrow < 0 || diag_wrow >= msg_rows) return BASQ_EINVAL;
    if (diag_tail_row < 0 || diag_tail_row >= msg_rows || n_tail_diag < 0 || n_tail_diag > S) return BASQ_EINVAL;
    if (diagU && (n_diag > ld_diag || n_tail_diag > ld_diag)) return BASQ_EINVAL;
    const int tot = (q + 1) * S;
    hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, parts,
                       n_parts, msg_rows, q, S, diagU, (long long)ld_diag, n_diag, diag_noise, diag_wrow, diag_tail_row,
                       n_tail_diag, XcarT, tot_out, (const long long*)geo);
    BASQ_CHECK_LAUNCH();
    return BASQ_OK;
}

int basq_finalize_f64(const double* parts, int32_t n_parts, int32_t msg_rows, int32_t q, int32_t S,
                      const double* diagU, int64_t ld_diag, int32_t n_diag, double diag_noise, int32_t diag_wrow,
                      int32_t diag_tail_row, int32_t n_tail_diag, double* XcarT, double* tot_out, void* stream) {
    return basq_finalize_geo_f64(parts, n_parts, msg_rows, q, S, diagU, ld_diag, n_diag, diag_noise, diag_wrow,
                                 diag_tail_row, n_tail_diag, nullptr, XcarT, tot_out, stream);
}

// tail weights of a descriptor-driven round: out[k] = mu * wx of tail point k (positions n_full + k), zero beyond the tail
__global__ void tail_weights_geo_kernel(const double* __restrict__ mu, const double* __restrict__ wx,
                                        const long long* __restrict__ geo, int S, double* __restrict__ out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= S) return;
    const long long n_full = geo[1], n_tail = geo[5], off = geo[6], Rl = geo[7];
    double v = 0.0;
    const long long p = n_full + k - off;                     // local index of tail point k (this rank may hold only some)
    if (k < n_tail && p >= 0 && p < Rl) v = wx ? mu[p] * wx[p] : mu[p];
    out[k] = v;
}

int basq_tail_weights_geo_f64(const double* mu, const double* wx, const int64_t* geo, int32_t S, double* out, void* stream) {
    if (!mu || !geo || !out || S < 1) return BASQ_EINVAL;
    hipLaunchKernelGGL(tail_weights_geo_kernel, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mu, wx,
                       (const long long*)geo, S, out);
    BASQ_CHECK_LAUNCH();
    return BASQ_OK;
}

// workspace (doubles) of the cluster kernels for an [s, M] reduction: 16 counter/flag words + the message ring
// (ring of 2 W slots of tagged 16-byte granules; the bidiagonalisation's 2 x (NCU + 1) messages are the smaller user)
#ifndef BASQ_CLUSTER_NCU
#define BASQ_CLUSTER_NCU 8       // work-groups of a cluster (8 waves each), all dealt to ONE XCD; rows per wave = 256 / (8 NCU).
                                 // 8 vs 4 at 200 x 400: null space 779 vs 845 us, elimination 521 vs 564 (profiles/r04_u_*)
#endif
#define BASQ_CLUSTER_NR (32 / BASQ_CLUSTER_NCU)
// Work-groups b, b + 8, b + 16, ... of a launch are dealt to one XCD (observed, not promised): a cluster uses every 8th
// work-group of its grid.  BASQ_CLUSTER_SPREAD=1 (tests) uses consecutive work-groups instead -- eight different XCDs -- so
// that the write-through path the kernels fall back to when their members do NOT share an XCD is exercised on purpose.
static int cluster_stride() {
    static const int spread = [] { const char* e = getenv("BASQ_CLUSTER_SPREAD"); return (e && atoi(e) > 0) ? 1 : 0; }();
    return spread ? 1 : 8;
}
static inline size_t bidiag_ws_doubles(int nv, int ncu) { return 16 + 2 * (size_t)(2 * (ncu + 1)) * (nv * 64 + 8); }
static inline size_t gring_ws_doubles(int M, int nrows) { return 16 + 2 * (size_t)nrows * (((M + 63) / 64) * 64 + 4); }
static inline size_t cluster_ws_doubles(int nv, int ncu) { return 16 + 2 * (size_t)(2 * BASQ_WPG * ncu) * (nv * 64 + 8); }

#ifndef BASQ_CAR_CLUSTER
#define BASQ_CAR_CLUSTER 1      // 1: cluster kernels where the null vectors do not fit one CU's LDS; 2: also where they do
#endif                          // (A/B: 172 vs 175 us at M = 200, slower below); 0: never

int64_t basq_reduction_ws_doubles(int32_t s, int32_t M) {
    if (s < 1 || M <= s || M > 1024) return 0;
    size_t need = 0;
    // clusters of BASQ_CLUSTER_NCU work-groups (elimination ring: 2 W slots; bidiagonalisation: 2 x (NCU + 1) messages -- the ring is the larger)
    if (M > 256 && M <= 512 && (s <= 256 || (M - s) <= 256)) need = cluster_ws_doubles(8, BASQ_CLUSTER_NCU);
    // the elimination's global ring (car_eliminate_gring_kernel): one slot of tagged granules per pivot
    if (M > 256 && M <= 448 && (M - s) <= 256) {
        const size_t ring = gring_ws_doubles(M, M - s);
        if (ring > need) need = ring;
    }
    return (int64_t)need;
}

int basq_car_eliminate_f64(double* PhiT, double* mu, int32_t M, int32_t s, int32_t* keep_rank, int32_t* kept,
                           double* w_star, int32_t* info, double* ws, void* stream) {
    if (!PhiT || !mu || !keep_rank || !kept || !w_star || !info || M < 1 || M > 1024 || s < 1 || s > M)
        return BASQ_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int nrows = M - s;
    {
        // one CU, null vectors in registers, handed over in blocks of NR rows (BASQ_CAR_RING=0: the LDS-resident kernel)
        static const int ring_env = [] { const char* e = getenv("BASQ_CAR_RING"); return e ? atoi(e) : 1; }();
        const size_t ring_lds = (size_t)nrows * (M + 4) * sizeof(double);
        if (ring_env && nrows >= 1 && M <= 256 && nrows <= BASQ_RING_WPG * BASQ_RING_NR && ring_lds <= 163328) {   // 163840 B per CU - static LDS
            if (nrows <= 16 * 4) {
                if (hipFuncSetAttribute((const void*)car_eliminate_ring_kernel<4, 16>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)ring_lds) != hipSuccess)
                    return BASQ_ELAUNCH;
                hipLaunchKernelGGL((car_eliminate_ring_kernel<4, 16>), dim3(1), dim3(1024), ring_lds, st, PhiT, mu, M, s, keep_rank,
                                   kept, w_star, info);
            } else {
                if (hipFuncSetAttribute((const void*)car_eliminate_ring_kernel<BASQ_RING_NR, BASQ_RING_WPG>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)ring_lds) != hipSuccess)
                    return BASQ_ELAUNCH;
                hipLaunchKernelGGL((car_eliminate_ring_kernel<BASQ_RING_NR, BASQ_RING_WPG>), dim3(1), dim3(BASQ_RING_WPG * 64), ring_lds, st,
                                   PhiT, mu, M, s, keep_rank, kept, w_star, info);
            }
            BASQ_CHECK_LAUNCH();
            return BASQ_OK;
        }
    }
#if BASQ_CAR_CLUSTER
    constexpr int T = BASQ_WPG * 64;
    const bool fits_lds = (size_t)nrows * M * sizeof(double) <= 162560;
    if (nrows >= 1 && M <= 256 && nrows <= BASQ_WPG * 14 && (BASQ_CAR_CLUSTER == 2 || !fits_lds)) {   // one CU, rows in registers
        if (nrows <= BASQ_WPG * 4)
            hipLaunchKernelGGL((car_eliminate_cluster_kernel<4, 4, 1>), dim3(1), dim3(T), 0, st, PhiT, mu, M, s,
                               keep_rank, kept, w_star, info, (double*)nullptr, 1);
        else
            hipLaunchKernelGGL((car_eliminate_cluster_kernel<4, 14, 1>), dim3(1), dim3(T), 0, st, PhiT, mu, M, s,
                               keep_rank, kept, w_star, info, (double*)nullptr, 1);
        BASQ_CHECK_LAUNCH();
        return BASQ_OK;
    }
    {
        // rows in registers over several work-groups, pivots through a global ring (BASQ_CAR_GRING=0: the cluster kernel)
        static const int gring_env = [] { const char* e = getenv("BASQ_CAR_GRING"); return e ? atoi(e) : 1; }();
        if (gring_env && nrows >= 1 && M > 256 && M <= 448 && nrows <= 256 && ws) {
            constexpr int RPG = BASQ_GRING_NR * BASQ_GRING_WPG;               // rows per work-group (8 waves x 4 rows)
            const int nv = (M + 63) / 64, n_groups = (nrows + RPG - 1) / RPG;
            if (hipMemsetAsync(ws, 0, gring_ws_doubles(M, nrows) * sizeof(double), st) != hipSuccess) return BASQ_ELAUNCH;
            const dim3 grid((unsigned)(n_groups * cluster_stride()));
#define BASQ_GRING_LAUNCH(NVV)                                                                                              \
    hipLaunchKernelGGL((car_eliminate_gring_kernel<NVV, BASQ_GRING_NR, BASQ_GRING_WPG>), grid, dim3(BASQ_GRING_WPG * 64), 0, st, \
                       PhiT, mu, M, s, keep_rank, kept, w_star, info, ws, n_groups, cluster_stride())
            if (nv == 5) BASQ_GRING_LAUNCH(5);
            else if (nv == 6) BASQ_GRING_LAUNCH(6);
            else BASQ_GRING_LAUNCH(7);
#undef BASQ_GRING_LAUNCH
            BASQ_CHECK_LAUNCH();
            return BASQ_OK;
        }
    }
    if (nrows >= 1 && !fits_lds && M <= 512 && nrows <= 4 * BASQ_WPG * 8 && ws) {   // cluster of BASQ_CLUSTER_NCU CUs (n = 200: M = 400)
        // every granule word zeroed: tags are the step numbers of THIS launch
        if (hipMemsetAsync(ws, 0, cluster_ws_doubles(8, BASQ_CLUSTER_NCU) * sizeof(double), st) != hipSuccess) return BASQ_ELAUNCH;
        hipLaunchKernelGGL((car_eliminate_cluster_kernel<8, BASQ_CLUSTER_NR, BASQ_CLUSTER_NCU>), dim3(BASQ_CLUSTER_NCU * cluster_stride()), dim3(T), 0, st, PhiT, mu, M, s,
                           keep_rank, kept, w_star, info, ws, cluster_stride());
        BASQ_CHECK_LAUNCH();
        return BASQ_OK;
    }
#endif
    const size_t lds = (size_t)(M - s) * M * sizeof(double);
    if (s < M && lds <= 162560) {      // 163840 B per CU minus the kernel's static LDS
        if (hipFuncSetAttribute((const void*)car_eliminate_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return BASQ_ELAUNCH;
        // 1024 threads: A/B-measured 4.8 ms per batch vs 5.6 (512) and 8.2 (256) -- the rank-1 updates dominate
        const int nthreads = (M <= BASQ_CAR_THREADS) ? BASQ_CAR_THREADS : ((M <= 512) ? 512 : 1024);
        hipLaunchKernelGGL(car_eliminate_lds_kernel, dim3(1), dim3(nthreads), lds, st, PhiT, mu, M, s,
                           keep_rank, kept, w_star, info);
        BASQ_CHECK_LAUNCH();
        return BASQ_OK;
    }
    hipLaunchKernelGGL(car_eliminate_kernel, dim3(1), dim3(1024), 0, st, PhiT, mu, M, s, keep_rank,
                       kept, w_star, info);
    BASQ_CHECK_LAUNCH();
    return BASQ_OK;
}

#ifndef BASQ_NS_CLUSTER
#define BASQ_NS_CLUSTER 1       // 1: cluster kernels where one CU cannot hold the matrix (M > 256); 2: also for the one-CU
#endif                          // shapes (A/B: measured 336 vs 286 us at 100 x 200 -- the 16-wave kernel stays); 0: never
int basq_nullspace_f64(const double* X, int32_t s, int32_t M, double* V, double* tau, double* PhiT, double* ws,
                       int32_t* info, void* stream) {
    if (!X || !V || !tau || !PhiT || s < 1 || M <= s || M > 1024) return BASQ_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (info && hipMemsetAsync(info, 0, sizeof(int32_t), st) != hipSuccess) return BASQ_ELAUNCH;
    const size_t LDS_MAX = 163840 - 64;                         // per-CU LDS minus the kernel's static scalar
    int rc;
    constexpr int T = BASQ_WPG * 64;
    if (BASQ_NS_CLUSTER == 2 && M <= 256 && s <= BASQ_WPG * 14) {   // one CU, 8 waves (A/B builds)
        if (s <= BASQ_WPG * 4) hipLaunchKernelGGL((bidiag_cluster_kernel<4, 4, 1>), dim3(1), dim3(T), 0, st, X, s, M, V, tau, (double*)nullptr, 1, info);
        else hipLaunchKernelGGL((bidiag_cluster_kernel<4, 14, 1>), dim3(1), dim3(T), 0, st, X, s, M, V, tau, (double*)nullptr, 1, info);
        rc = (hipGetLastError() == hipSuccess) ? BASQ_OK : BASQ_ELAUNCH;
    } else if (BASQ_NS_CLUSTER && !(M <= 256 && s <= 112) && M <= 510 && s <= 4 * BASQ_WPG * 8 && ws) {   // cluster of BASQ_CLUSTER_NCU CUs
        // every granule word zeroed: tags are the step numbers of THIS launch (16 + 2 x (4 + 1) messages of 520 granules)
        if (hipMemsetAsync(ws, 0, bidiag_ws_doubles(8, BASQ_CLUSTER_NCU) * sizeof(double), st) != hipSuccess) return BASQ_ELAUNCH;
        hipLaunchKernelGGL((bidiag_cluster_kernel<8, BASQ_CLUSTER_NR, BASQ_CLUSTER_NCU>), dim3(BASQ_CLUSTER_NCU * cluster_stride()), dim3(T), 0, st, X, s, M, V, tau, ws, cluster_stride(), info);
        rc = (hipGetLastError() == hipSuccess) ? BASQ_OK : BASQ_ELAUNCH;
    } else if (M <= 256 && s <= 112) {                          // whole matrix in registers (16 waves)
        if (s <= 32) hipLaunchKernelGGL((bidiag_reflectors_reg_kernel<4, 2>), dim3(1), dim3(1024), 0, st, X, s, M, V, tau);
        else if (s <= 64) hipLaunchKernelGGL((bidiag_reflectors_reg_kernel<4, 4>), dim3(1), dim3(1024), 0, st, X, s, M, V, tau);
        else hipLaunchKernelGGL((bidiag_reflectors_reg_kernel<4, 7>), dim3(1), dim3(1024), 0, st, X, s, M, V, tau);
        rc = (hipGetLastError() == hipSuccess) ? BASQ_OK : BASQ_ELAUNCH;
    } else if (M <= 256) {
        const size_t fixed = (size_t)(2 * M + s + 16 * M) * sizeof(double);
        const size_t rows = (size_t)(s > 32 ? s - 32 : 0) * M * sizeof(double);
        if (fixed + rows <= LDS_MAX) rc = launch_bidiag<4, 2, 16, true>(X, s, M, V, tau, fixed + rows, st);
        else rc = launch_bidiag<4, 2, 16, false>(X, s, M, V, tau, fixed, st);
    } else if (M <= 512) {
        rc = launch_bidiag<8, 0, 16, false>(X, s, M, V, tau, (size_t)(2 * M + s + 16 * M) * sizeof(double), st);
    } else {
        rc = launch_bidiag<16, 0, 8, false>(X, s, M, V, tau, (size_t)(2 * M + s + 8 * M) * sizeof(double), st);
    }
    if (rc != BASQ_OK) return rc;
    const int nvec = M - s;
    // (a form with 16 lanes per null vector was the faster one for M > 256 while four reflector rows were in flight; with
    // twelve, the 64-lane form wins there too: 68 vs 167 us at 200 x 400 -- profiles/r02_n_nullspace_apply_200x400.txt)
    const dim3 grid((nvec + 3) / 4), block(256);
    if (M <= 256) hipLaunchKernelGGL(nullspace_apply_kernel<4>, grid, block, 0, st, V, tau, s, M, PhiT);
    else if (M <= 512) hipLaunchKernelGGL(nullspace_apply_kernel<8>, grid, block, 0, st, V, tau, s, M, PhiT);
    else hipLaunchKernelGGL(nullspace_apply_kernel<16>, grid, block, 0, st, V, tau, s, M, PhiT);
    BASQ_CHECK_LAUNCH();
    return BASQ_OK;
}

int basq_reweight_compact_f64(const double* cand, const double* mu, const int64_t* gid, const double* wx,
                              int64_t Rl, int64_t off, int64_t n_full, int32_t S, int32_t kp,
                              const int32_t* keep_rank, const double* w_star, const double* tot, int32_t n_keep,
                              int64_t new_off, double* cand_out, double* mu_out, int64_t* gid_out, double* wx_out,
                              void* stream) {
    if (!cand || !mu || !gid || !keep_rank || !w_star || !tot || !cand_out || !mu_out || !gid_out) return BASQ_EINVAL;
    if (Rl < 0 || S < 1 || kp < 1 || n_keep < 0 || (wx && !wx_out)) return BASQ_EINVAL;
    if (Rl == 0) return BASQ_OK;
    const long long nt = (long long)Rl * kp;
    hipLaunchKernelGGL(reweight_compact_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       cand, mu, (const long long*)gid, wx, (long long)Rl, (long long)off, (long long)n_full, S, kp,
                       keep_rank, w_star, tot, n_keep, (long long)new_off, cand_out, mu_out, (long long*)gid_out,
                       wx_out);
    BASQ_CHECK_LAUNCH();
    return BASQ_OK;
}

int basq_blocksum_geo_f64(const basq_kernel_spec* spec, const double* nys, int32_t m, const double* cand,
                          const double* mu, const double* wx, const int64_t* geo, int32_t geo_mode, int32_t S,
                          int32_t n_chunks, int32_t class_mod, int32_t class0, double* Xpart, double* totpart,
                          void* stream) {
    if (!spec_ok(spec) || !nys || !cand || !mu || !Xpart || !totpart || !geo) return BASQ_EINVAL;
    if (m < 1 || S < 1 || n_chunks < 1 || geo_mode < 1 || geo_mode > 4) return BASQ_EINVAL;
    if (class_mod < 0 || class0 < 0 || (class_mod > 0 && class0 + n_chunks > class_mod)) return BASQ_EINVAL;
    if (class_mod > 0 && geo_mode != 1) return BASQ_EINVAL;       // residue classes cover the regular region only
    BlocksumArgs A;
    A.nys = nys; A.cand = cand; A.mu = mu; A.wx = wx; A.Xpart = Xpart; A.totpart = totpart;
    A.Rl = 0; A.off = 0; A.n_full = 0; A.blk_lo = 0; A.blk_hi = 0; A.blk_per_chunk = 1;   // set on the device
    A.m = m; A.S = S; A.n_chunks = n_chunks;
    A.class_mod = class_mod; A.class0 = class0;
    A.geo = (const long long*)geo; A.geo_mode = geo_mode;
    A.n_stiles = (S + 15) / 16;
    return dispatch_blocksum(basq_kp(spec->d) / 4, spec->family, A, (hipStream_t)stream, blocksum_exp_scheme(spec));
}

int basq_reweight_compact_geo_f64(const double* cand, const double* mu, const int64_t* gid, const double* wx,
                                  const int64_t* geo, const int64_t* geo_next, const int32_t* info, int64_t R_max,
                                  int32_t S, int32_t kp,
                                  const int32_t* keep_rank, const double* w_star, const double* tot, int64_t out_rows,
                                  int32_t expect_keep, double* cand_out, double* mu_out, int64_t* gid_out, double* wx_out,
                                  void* stream) {
    if (!cand || !mu || !gid || !geo || !geo_next || !info || !keep_rank || !w_star || !tot || !cand_out || !mu_out ||
        !gid_out)
        return BASQ_EINVAL;
    if (R_max < 1 || S < 1 || kp < 1 || out_rows < 1 || (wx && !wx_out)) return BASQ_EINVAL;
    const long long nt = (long long)R_max * kp;
    hipLaunchKernelGGL(reweight_compact_geo_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       cand, mu, (const long long*)gid, wx, (const long long*)geo, (const long long*)geo_next, info, S, kp,
                       keep_rank, w_star, tot, (long long)out_rows, expect_keep, cand_out, mu_out, (long long*)gid_out, wx_out);
    BASQ_CHECK_LAUNCH();
    return BASQ_OK;
}

int basq_round_next_i64(const int64_t* geo, const int32_t* info, const int32_t* keep_rank, int32_t S, int32_t class_mode,
                        int32_t expect_half, int64_t* geo_next, void* stream) {
    if (!geo || !info || !keep_rank || !geo_next || S < 1) return BASQ_EINVAL;
    hipLaunchKernelGGL(round_next_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const long long*)geo, info, keep_rank,
                       S, class_mode, expect_half, (long long*)geo_next);
    BASQ_CHECK_LAUNCH();
    return BASQ_OK;
}

int basq_init_state_f64(double* mu, int64_t* gid, int64_t Rl, int64_t gid0, int64_t n_total, void* stream) {
    if (!mu || !gid || Rl < 0 || n_total < 1) return BASQ_EINVAL;
    if (Rl == 0) return BASQ_OK;
    // torch.ones(N) / N (BASQ/_rchq.py:53): one correctly rounded division
    hipLaunchKernelGGL(init_state_kernel, dim3((unsigned)((Rl + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mu,
                       (long long*)gid, (long long)Rl, (long long)gid0, 1.0 / (double)n_total);
    BASQ_CHECK_LAUNCH();
    return BASQ_OK;
}

int basq_box_muller_f64(const double* u, int64_t n, const double* u_tail, double* out, void* stream) {
    if (!u || !out || n < 16 || ((n % 16 != 0) != (u_tail != nullptr))) return BASQ_EINVAL;
    const long long nblk = n / 16;
    hipLaunchKernelGGL(box_muller_kernel, dim3((unsigned)((nblk * 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, u,
                       nblk, out);
    BASQ_CHECK_LAUNCH();
    if (u_tail) {   // torch recomputes the LAST 16 values from fresh uniforms; launched second: it overwrites
        hipLaunchKernelGGL(box_muller_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, u_tail, 1LL, out + (n - 16));
        BASQ_CHECK_LAUNCH();
    }
    return BASQ_OK;
}

int basq_dense_blocksum_f64(const double* C, int32_t m, int64_t nc, int64_t ldc, const double* mu, int64_t pg0,
                            int64_t n_full, int32_t S, double scale, int32_t square, double* E, double* tot, void* stream) {
    if (!C || !mu || !E || m < 1 || nc < 0 || ldc < nc || pg0 < 0 || n_full < 0 || S < 1 || n_full % S != 0)
        return BASQ_EINVAL;
    if (nc == 0) return BASQ_OK;
    if (S % 2 == 0 && pg0 % 2 == 0 && S >= 4 && S <= 2048 && nc >= 4 * (int64_t)S) {
        // 16 bytes per lane: set pairs x position slices (dense_blocksum_pairs_kernel).  Slices: as many as keep the whole
        // grid resident at once (20 waves per CU at this kernel's 84 registers): the launch then has no second, partly
        // filled round of work-groups (1250 work-groups of 7 waves ran as 1.6 rounds: 81 % of the time useful).
        constexpr int JR8 = 8;
        const int half = S / 2;
        const long long grid_n = (m + JR8 - 1) / JR8;
        const long long nblocks = (nc + S - 1) / S;
        int NS = 1;
        for (int cand_ns = 2; cand_ns <= 16; ++cand_ns) {
            const long long waves = grid_n * ((half * cand_ns + 63) / 64);
            if (half * cand_ns > 1024 || waves > 256LL * 20 || (long long)cand_ns * 2 > nblocks) break;
            NS = cand_ns;
        }
        static const int ns_env = [] { const char* e = getenv("BASQ_DBS_NS"); return e ? atoi(e) : 0; }();   // A/B knob
        if (ns_env > 0 && half * ns_env <= 1024 && (long long)ns_env * 2 <= nblocks) NS = ns_env;
        const int nthr = half * NS;
        size_t lds = (NS > 1) ? (size_t)NS * (JR8 + 1) * S * sizeof(double) : 0;
        const size_t lds_tail = (size_t)((nthr + 63) / 64) * (JR8 + 1) * sizeof(double);
        if (lds < lds_tail) lds = lds_tail;
        if (nthr <= 1024 && lds <= 160 * 1024 - 512) {
            const dim3 grid8((unsigned)grid_n), block8((unsigned)nthr);
            // (A/B knob BASQ_DBS_NT=0: plain loads.  Measured on 1-GB chunks, S = 400: 6.05 vs 5.19 TB/s back to back, 4.88 vs
            //  4.21 TB/s when the chunk was written by an element-wise kernel just before, as inside a batch --
            //  profiles/r06_f_dense_blocksum_nontemporal_ab.txt)
            static const int nt_env = [] { const char* e = getenv("BASQ_DBS_NT"); return e ? atoi(e) : 1; }();
            // non-temporal 16-byte loads need 16-byte aligned pairs: even row stride, aligned base
            const bool nt = nt_env != 0 && ldc % 2 == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0;
            const void* fn = square ? (nt ? (const void*)dense_blocksum_pairs_kernel<JR8, true, true>
                                          : (const void*)dense_blocksum_pairs_kernel<JR8, true, false>)
                                    : (nt ? (const void*)dense_blocksum_pairs_kernel<JR8, false, true>
                                          : (const void*)dense_blocksum_pairs_kernel<JR8, false, false>);
            if (lds > 64 * 1024 &&
                hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return BASQ_ELAUNCH;
#define BASQ_DBS_LAUNCH(SQv, NTv)                                                                                         \
    hipLaunchKernelGGL((dense_blocksum_pairs_kernel<JR8, SQv, NTv>), grid8, block8, lds, (hipStream_t)stream, C, m,         \
                       (long long)nc, (long long)ldc, mu, (long long)pg0, (long long)n_full, S, NS, scale, E, tot)
            if (square) { if (nt) BASQ_DBS_LAUNCH(true, true); else BASQ_DBS_LAUNCH(true, false); }
            else { if (nt) BASQ_DBS_LAUNCH(false, true); else BASQ_DBS_LAUNCH(false, false); }
#undef BASQ_DBS_LAUNCH
            BASQ_CHECK_LAUNCH();
            return BASQ_OK;
        }
    }
    constexpr int JR = 4;
    const dim3 grid((unsigned)((m + JR - 1) / JR)), block(256);
    if (square)
        hipLaunchKernelGGL((dense_blocksum_kernel<JR, true>), grid, block, 0, (hipStream_t)stream, C, m, (long long)nc,
                           (long long)ldc, mu, (long long)pg0, (long long)n_full, S, scale, E, tot);
    else
        hipLaunchKernelGGL((dense_blocksum_kernel<JR, false>), grid, block, 0, (hipStream_t)stream, C, m, (long long)nc,
                           (long long)ldc, mu, (long long)pg0, (long long)n_full, S, scale, E, tot);
    BASQ_CHECK_LAUNCH();
    return BASQ_OK;
}

int basq_chol_inv_f64(double* G, int32_t q, double* W, int32_t* info, double rel_tol, void* stream) {
    if (!G || !info || q < 1 || q > 1024 || !(rel_tol >= 0.0)) return BASQ_EINVAL;
    if (!W) {                                                   // factor only
        const size_t ldsp = ((size_t)q * (q + 1) / 2 + q) * sizeof(double);
        if (ldsp > 163840 - 256) return BASQ_EUNSUPPORTED;
        if (hipFuncSetAttribute((const void*)chol_packed_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)ldsp) != hipSuccess)
            return BASQ_ELAUNCH;
        hipLaunchKernelGGL(chol_packed_lds_kernel, dim3(1), dim3(1024), ldsp, (hipStream_t)stream, G, q, info, rel_tol);
        BASQ_CHECK_LAUNCH();
        return BASQ_OK;
    }
    const size_t sq = (size_t)q * (q | 1);
    const size_t qpad = (size_t)((q + 1) & ~1);
    const size_t lds1 = (sq + qpad) * sizeof(double), lds2 = (2 * sq + qpad) * sizeof(double);
    const size_t LDS_MAX = 163840 - 256;                        // per-CU LDS minus the kernel's static part
    if (lds1 <= LDS_MAX) {
        const int blocked = (lds2 <= LDS_MAX && q >= 8) ? 1 : 0;
        const size_t lds = blocked ? lds2 : lds1;
        if (hipFuncSetAttribute((const void*)chol_inv_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess)
            return BASQ_ELAUNCH;
        hipLaunchKernelGGL(chol_inv_lds_kernel, dim3(1), dim3(BASQ_CHOL_THREADS), lds, (hipStream_t)stream, G, q, W, info,
                           rel_tol, blocked);
        BASQ_CHECK_LAUNCH();
        return BASQ_OK;
    }
    hipLaunchKernelGGL(chol_inv_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, G, q, W, info, rel_tol);
    BASQ_CHECK_LAUNCH();
    return BASQ_OK;
}

int basq_chol_factor_f64(double* G, int32_t q, int32_t* info, double rel_tol, void* stream) {
    if (!G || !info || q < 1 || !(rel_tol >= 0.0)) return BASQ_EINVAL;
    const size_t ldsp = (size_t)q * (q + 1) / 2 * sizeof(double);
    if (ldsp > 163840 - 512) return BASQ_EUNSUPPORTED;           // q <= 200
    // Every thread factors the 8 x 8 diagonal block of a panel redundantly: with 16 waves that serial part runs four times
    // per SIMD, and the trailing update (at most 1176 tiles at q = 200) does not need them: 512 threads instead of 1024:
    // q = 99 102.9 -> 68.8 us (256 threads: 79.1), q = 199 255.5 -> 170.4 us (profiles/r02_m_chol_threads.txt).
    if (hipFuncSetAttribute((const void*)chol_factor_panel_kernel<512>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)ldsp) != hipSuccess)
        return BASQ_ELAUNCH;
    hipLaunchKernelGGL(chol_factor_panel_kernel<512>, dim3(1), dim3(512), ldsp, (hipStream_t)stream, G, q, info, rel_tol);
    BASQ_CHECK_LAUNCH();
    return BASQ_OK;
}

int basq_trsm_rows_f64(const double* X, int64_t ldx, int64_t rows, int32_t q, const double* L, double* Q, int64_t ldq,
                       void* stream) {
    if (!X || !L || !Q || rows < 0 || q < 1 || ldx < q || ldq < q) return BASQ_EINVAL;
    if (rows == 0) return BASQ_OK;
    const size_t lds = (size_t)64 * (q | 1) * sizeof(double);
    if (lds > 163840 - 256) return BASQ_EUNSUPPORTED;            // q <= 318
    const size_t lds_l = lds + (size_t)q * q * sizeof(double);   // with the factor in LDS as well: q <= 112
    const dim3 grid((unsigned)((rows + 63) / 64));
    if (lds_l <= 163840 - 256) {
        if (hipFuncSetAttribute((const void*)trsm_rows_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_l) !=
            hipSuccess)
            return BASQ_ELAUNCH;
        hipLaunchKernelGGL(trsm_rows_kernel<true>, grid, dim3(512), lds_l, (hipStream_t)stream, X, (long long)ldx,
                           (long long)rows, q, L, Q, (long long)ldq);
    } else {
        if (hipFuncSetAttribute((const void*)trsm_rows_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess)
            return BASQ_ELAUNCH;
        hipLaunchKernelGGL(trsm_rows_kernel<false>, grid, dim3(512), lds, (hipStream_t)stream, X, (long long)ldx,
                           (long long)rows, q, L, Q, (long long)ldq);
    }
    BASQ_CHECK_LAUNCH();
    return BASQ_OK;
}

int basq_cholqr_f64(double* G, int32_t q, int32_t* info, double rel_tol, const double* X, int64_t ldx, int64_t rows,
                    double* Q, int64_t ldq, void* stream) {
    if (!G || !info || !X || !Q || q < 1 || rows < 0 || ldx < q || ldq < q || !(rel_tol >= 0.0)) return BASQ_EINVAL;
    const size_t tri = (size_t)q * (q + 1) / 2 * sizeof(double);                        // the factor's packed triangle
    const size_t lds_full = ((size_t)64 * (q | 1) + (size_t)q * q) * sizeof(double);    // a solver's rows + its image of L
    const size_t lds_rows = ((size_t)64 * (q | 1) + (size_t)BASQ_CHOL_NB * q) * sizeof(double);   // ... + one row panel of L
    const bool full = lds_full <= 163840 - 512;                                       // q <= 112
    const size_t lds = full ? lds_full : (lds_rows > tri ? lds_rows : tri);
    if (lds > 163840 - 512 || rows > 64LL * 4096) return BASQ_EUNSUPPORTED;           // q <= 200; every work-group resident
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(info, 0, 2 * sizeof(int32_t), st) != hipSuccess) return BASQ_ELAUNCH;   // pivot flag | progress word
    const dim3 grid((unsigned)(1 + (rows + 63) / 64));
    if (full) {
        if (hipFuncSetAttribute((const void*)cholqr_fused_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return BASQ_ELAUNCH;
        hipLaunchKernelGGL(cholqr_fused_kernel<true>, grid, dim3(512), lds, st, G, q, info, rel_tol, X, (long long)ldx,
                           (long long)rows, Q, (long long)ldq);
    } else {
        if (hipFuncSetAttribute((const void*)cholqr_fused_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return BASQ_ELAUNCH;
        hipLaunchKernelGGL(cholqr_fused_kernel<false>, grid, dim3(512), lds, st, G, q, info, rel_tol, X, (long long)ldx,
                           (long long)rows, Q, (long long)ldq);
    }
    BASQ_CHECK_LAUNCH();
    return BASQ_OK;
}

int basq_skinny_gemm_f64(const double* A, int64_t lda, int32_t trans, int32_t M, int32_t K, const double* B, int64_t ldb,
                         int32_t N, int32_t ksplit, double* work, double* C, void* stream) {
    if (!A || !B || !C || M < 1 || N < 1 || K < 1 || ksplit < 1 || ldb < N) return BASQ_EINVAL;
    if (lda < (trans ? M : K)) return BASQ_EINVAL;
    if (N > 208 || ldb > (1LL << 24)) return BASQ_EUNSUPPORTED;   // (the kernel keeps 15 ldb + 16 doubles as a 32-bit byte offset)
    hipStream_t st = (hipStream_t)stream;
    int kslice = (K + ksplit - 1) / ksplit;
    kslice = ((kslice + 15) / 16) * 16;                          // whole 16-k trips per slice
    const int nz = (K + kslice - 1) / kslice;
    if (nz > 1 && !work) return BASQ_EINVAL;
    double* out = (nz > 1) ? work : C;
    const long long cstride = (long long)M * N;
    dispatch_skinny(trans != 0, M, nz, 1, st, A, lda, 0LL, B, ldb, out, cstride, N, K, kslice);
    BASQ_CHECK_LAUNCH();
    if (nz > 1) {
        const long long n = cstride;
        hipLaunchKernelGGL(sum_parts_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, work, nz, n, C);
        BASQ_CHECK_LAUNCH();
    }
    return BASQ_OK;
}

int basq_gemm_f64(const double* A, int64_t lda, const double* B, int64_t ldb, double* C, int64_t ldc, int32_t M,
                  int32_t N, int32_t K, double alpha, void* stream) {
    if (!A || !B || !C || M < 1 || N < 1 || K < 1 || lda < K || ldb < N || ldc < N) return BASQ_EINVAL;
    dim3 grid((unsigned)((M + 63) / 64), (unsigned)((N + 63) / 64), 1);
    const int kslice = ((K + 3) / 4) * 4;
    hipLaunchKernelGGL((gemm_kernel<4>), grid, dim3(256), 0, (hipStream_t)stream, A, (long long)lda, 1LL, B, (long long)ldb,
                       0LL, 1, C, (long long)ldc, 0LL, M, N, K, kslice, alpha, 1, 0LL);
    BASQ_CHECK_LAUNCH();
    return BASQ_OK;
}

}  // extern "C"
