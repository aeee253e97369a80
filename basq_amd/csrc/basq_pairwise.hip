// basq_pairwise.hip -- pairwise-kernel family: packing, fused block sums (+ squared covariances), Gram / mat-vec,
// dense block sums, and their C-ABI entries (include/basq_hip.h).
//
// Hot path: kernel recombination of ma921/BASQ (BASQ/_rchq.py).  Everything is float64 (SURVEY §8c:
// the reference's index selection is only well-posed in fp64).  Wave = 64 lanes; the pairwise
// exponent arguments are produced on the f64 matrix cores (v_mfma_f64_16x16x4_f64) from packed
// operands, the transcendental epilogue runs on the fp64 VALU, which is the binding unit.
//
// MFMA f64 16x16x4 lane maps (cdna_hip_programming.md §3):  lane l, c = l & 15, g = l >> 4
//     A operand: A[row = c][k = g]      B operand: B[k = g][col = c]
//     C/D:       D[reg r] = D[row = g + 4 r][col = c]
#include "basq_common.hpp"
#include "exp_coeffs.inc"

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

// Scaled-argument form of the two table schemes (round 5; the block sums): the Nystrom fragments are multiplied by N/ln2
// once per launch (they stay in registers), so the matrix instruction delivers y = x N/ln2 itself, and
//     ti = (int)y = -floor(|y|) = N n + j   (v_cvt_i32_f64 truncates),      f = |y| - floor(|y|)   (v_fract_f64, |.| a modifier)
// replace the magic-number round, the subtraction of the magic number and the FMA that forms r: exp(x) = 2^n T[j] 2^(-f/N) with a
// polynomial in f on [0, 1) (same interval width, same errors: 2.5e-14 / 1.7e-17) -- ONE fp64 instruction fewer per kernel value
// (11 instead of 12 VALU beside the 12 matrix lane-operations at d = 10).  Both halves read |y|: an argument that rounding left a hair
// above zero (coincident points: D = a.a + 2 h = +-1e-16 |a|^2) evaluates exp(-|x|), 2 |x| ~ 1e-15 away.  Below ~-3.6e5 (the
// conversion saturates; exp = 0 from -745 on) the result is an exact 0 as before.
#ifndef BASQ_BS_SCALED_ARG
#define BASQ_BS_SCALED_ARG 1       // A/B builds: -DBASQ_BS_SCALED_ARG=0 (the unscaled form above; same results to ~1e-15)
#endif
template <int XS>
struct ExpScale {
    static constexpr double value = (XS == 2) ? BASQ_4096_OVER_LN2 : BASQ_2048_OVER_LN2;    // N / ln2
};

// `arg_scale`: what takes the Matern kernels' r to y (-sqrt(5) N/ln2, -sqrt(3) N/ln2); unused for RBF
template <int XS>
__device__ __forceinline__ void expk_init_scaled(ExpK& k, double arg_scale) {
    k.k32 = vgpr_const(arg_scale);
    k.nhi = 0.0;
    k.magic = 0.0;
    if (XS == 2) {
        k.c3 = vgpr_const(BASQ_EXP_W2);
        k.c2 = vgpr_const(BASQ_EXP_W1);
        k.one = vgpr_const(BASQ_EXP_W0);
    } else {
        k.c3 = vgpr_const(BASQ_EXP_B3);
        k.c2 = vgpr_const(BASQ_EXP_B2);
        k.one = vgpr_const(BASQ_EXP_B1);
    }
}

template <int XS>
__device__ __forceinline__ double exp_scaled_k(double y, const ExpK& k, const double* tab) {
    const int ti = (int)y;                                              // N n + j (two's complement)
    const double f = __builtin_amdgcn_fract(__builtin_fabs(y));
    const double T = tab[ti & (ExpScheme<XS>::N - 1)];
    double w = __builtin_fma(k.c3, f, k.c2);
    w = __builtin_fma(w, f, k.one);
    const double e = (XS == 2) ? (T * w)                                // T (w0 + w1 f + w2 f^2)
                               : __builtin_fma(T * f, w, T);            // T (1 + f (b1 + b2 f + b3 f^2))
    return ldexp(e, ti >> ExpScheme<XS>::SHIFT);
}

// acc + W exp(x) for the block sums of the RBF kernel (round 6): the candidate's weight W is folded into the polynomial's
// coefficients ONCE per candidate fragment (three multiplications per 4 JT values: `ExpW`), and the table value takes the power of
// two BEFORE the product -- acc = fma(2^n T[j], W p(f), acc) -- so that the separate multiplications by the polynomial and by the
// weight become one FMA: 9 vector instructions per value instead of 10 (cvt, and, shift, fract, two FMAs, shift, ldexp, FMA).
// 2^n T[j] is exact (T in (1/2, 1]; n <= 0 down to the subnormals, where the value was an exact or rounded 0 before as well).
#ifndef BASQ_BS_WEIGHTED_POLY
#define BASQ_BS_WEIGHTED_POLY 1    // A/B builds: -DBASQ_BS_WEIGHTED_POLY=0 (round 5's sequence: e = ldexp(T p, n); acc += e W)
#endif
struct ExpW {
    double c2, c1, c0, w;          // W x the polynomial's coefficients; W itself (the 1e-17 scheme adds it)
};
template <int XS>
__device__ __forceinline__ ExpW expw_for(const ExpK&, double W) {
    // (the coefficients as LITERALS, not from ExpK's registers: three multiplications per candidate fragment can afford a scalar
    //  operand each, and the six registers ExpK holds them in are then free in this form of the kernel)
    ExpW e;
    e.c2 = ((XS == 2) ? BASQ_EXP_W2 : BASQ_EXP_B3) * W;
    e.c1 = ((XS == 2) ? BASQ_EXP_W1 : BASQ_EXP_B2) * W;
    e.c0 = ((XS == 2) ? BASQ_EXP_W0 : BASQ_EXP_B1) * W;
    e.w = W;
    return e;
}
template <int XS>
__device__ __forceinline__ double exp_scaled_wacc(double y, const ExpW& k, const double* tab, double acc) {
    const int ti = (int)y;                                              // N n + j (two's complement)
    const double f = __builtin_amdgcn_fract(__builtin_fabs(y));
    const double T = tab[ti & (ExpScheme<XS>::N - 1)];
    double p = __builtin_fma(k.c2, f, k.c1);
    p = __builtin_fma(p, f, k.c0);
    if (XS != 2) p = __builtin_fma(f, p, k.w);                          // W (1 + f (b1 + b2 f + b3 f^2))
    return __builtin_fma(ldexp(T, ti >> ExpScheme<XS>::SHIFT), p, acc);
}

// Kernel value from the block sums' scaled product: RBF: Dp = -1/2 |(x-y)/l|^2 N/ln2; Matern: Dp = |(x-y)/l|^2 (the fragments
// times -2: exact), k.k32 = -sqrt(5) N/ln2 resp. -sqrt(3) N/ln2.
template <int FAM, int XS>
__device__ __forceinline__ double kernel_from_scaled_k(double Dp, const ExpK& k, const double* tab) {
    if (FAM == BASQ_FAMILY_RBF) {
        return exp_scaled_k<XS>(Dp, k, tab);
    } else {
        const double r2 = fmax(Dp, 1e-30);                              // gpytorch: clamp_min(1e-30) before sqrt
        const double r = BASQ_FAST_SQRT ? sqrt_pos_fast(r2) : sqrt(r2);
        const double e = exp_scaled_k<XS>(r * k.k32, k, tab);
        if (FAM == BASQ_FAMILY_MATERN52)
            return __builtin_fma(5.0 / 3.0, r2, __builtin_fma(0x1.1e3779b97f4a8p+1, r, 1.0)) * e;     // (1 + sqrt(5) r + 5/3 r^2) e
        else
            return __builtin_fma(0x1.bb67ae8584caap+0, r, 1.0) * e;                                     // (1 + sqrt(3) r) e
    }
}

// Sum over the 16 lanes that share g = lane >> 4 (the 16 columns of an MFMA tile); fixed butterfly
// order, result in every lane of the group.
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
                                               // 5: the full blocks beyond the regular region, [reg_hi, n_full), one chunk per block (class mode)
};

// Candidate range of a descriptor-driven launch (wave-uniform scalar loads and arithmetic; the formulas of blocksum_impl).
// The descriptor carries the round's GLOBAL geometry and this rank's shard [off, off + Rl) of the live positions
// (geo[6], geo[7]; one rank: [0, R)); the launch covers the intersection of the shard with the mode's position range.
// -> the local index of the launch's first position (what per-candidate side arrays must be advanced by).
template <int KP>
__device__ __forceinline__ long long blocksum_apply_geo(BlocksumArgs& A) {
    const long long R = A.geo[0], n_full = A.geo[1], reg_hi = A.geo[2];
    const long long s_off = A.geo[6], s_end = A.geo[6] + A.geo[7];
    long long lo = 0, hi = R;
    if (A.geo_mode == 1) hi = reg_hi;
    else if (A.geo_mode == 2) lo = reg_hi;
    else if (A.geo_mode == 4) lo = n_full;
    else if (A.geo_mode == 5) { lo = reg_hi; hi = n_full; }
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
    return skip;
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
    if constexpr (FAM != BASQ_FAMILY_RBF && BASQ_BS_SCALED_ARG && BASQ_BS_WEIGHTED_POLY && BASQ_FAST_SQRT) {
        // Matern: the weight goes into the PREFACTOR's coefficients (W (1 + sqrt(5) r + 5/3 r^2) resp. W (1 + sqrt(3) r)), the table
        // value takes the power of two first, and acc = fma(2^n T p(f), W pref(r), acc): one multiplication fewer per value
        const double W = f.w;
        const double cw1 = ((FAM == BASQ_FAMILY_MATERN52) ? 0x1.1e3779b97f4a8p+1 : 0x1.bb67ae8584caap+0) * W;
        const double cw2 = (5.0 / 3.0) * W;
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
            d4 D = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) D = __builtin_amdgcn_mfma_f64_16x16x4f64(a[jt][kk], f.b[kk], D, 0, 0, 0);
#pragma unroll
            for (int r_ = 0; r_ < 4; ++r_) {
                const double r2 = fmax(D[r_], 1e-30);                    // gpytorch: clamp_min(1e-30) before sqrt
                const double r = sqrt_pos_fast(r2);
                const double y = r * ek.k32;
                const int ti = (int)y;
                const double fr = __builtin_amdgcn_fract(__builtin_fabs(y));
                const double T = tab[ti & (ExpScheme<XS>::N - 1)];
                double p = __builtin_fma(ek.c3, fr, ek.c2);
                p = __builtin_fma(p, fr, ek.one);
                const double Tn = ldexp(T, ti >> ExpScheme<XS>::SHIFT);
                const double e = (XS == 2) ? (Tn * p) : __builtin_fma(Tn * fr, p, Tn);
                const double pref = (FAM == BASQ_FAMILY_MATERN52) ? __builtin_fma(cw2, r2, __builtin_fma(cw1, r, W))
                                                                  : __builtin_fma(cw1, r, W);
                acc[jt][r_] = __builtin_fma(e, pref, acc[jt][r_]);
            }
        }
        return;
    }
    if constexpr (FAM == BASQ_FAMILY_RBF && BASQ_BS_SCALED_ARG && BASQ_BS_WEIGHTED_POLY) {
        const ExpW ew = expw_for<XS>(ek, f.w);                          // (a masked column has W = 0: its values add exact zeros)
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
            d4 D = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) D = __builtin_amdgcn_mfma_f64_16x16x4f64(a[jt][kk], f.b[kk], D, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[jt][r] = exp_scaled_wacc<XS>(D[r], ew, tab, acc[jt][r]);
        }
        return;
    }
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) {
        d4 D = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) D = __builtin_amdgcn_mfma_f64_16x16x4f64(a[jt][kk], f.b[kk], D, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double kv = BASQ_BS_SCALED_ARG ? kernel_from_scaled_k<FAM, XS>(D[r], ek, tab)
                                                 : kernel_from_arg_k<FAM, XS>(D[r], ek, tab);
            acc[jt][r] = __builtin_fma(kv, f.w, acc[jt][r]);
        }
    }
}

// Blocks of prefetch distance for the candidate rows of the block sums: 2 while the third fragment keeps the kernel at
// three waves per SIMD (<= 170 registers: KP <= 32), 1 for the wider rows (KK = 9, 10: 214-222 registers with a third
// fragment).  Measured at the headline shape: 6.36 -> 6.20 ms per 14-class launch (profiles/r04_k_blocksum_prefetch2.txt).
#ifndef BASQ_BS_PREFETCH
#define BASQ_BS_PREFETCH 0      // 0 = by KK (A/B builds: -DBASQ_BS_PREFETCH=1 or 2)
#endif
#define BASQ_BS_PF_FOR(KK) (BASQ_BS_PREFETCH ? BASQ_BS_PREFETCH : ((KK) <= 8 ? 2 : 1))
// Round 6: the RBF kernel at d = 7..10 WITHOUT per-candidate kernel weights (no WSABI warp: the headline shape) is an instantiation of
// its own -- `NOWX`: the weight pointer is a compile-time null (one register pair per candidate fragment less, no select / product
// per fragment) -- and, with the polynomial's coefficients as literals (`expw_for`), it needs 128 registers where the general form
// needs 161: FOUR waves per SIMD at the same prefetch distance, no scratch.  6.93 -> 6.56 ms per 1e10 pairs, same box, alternating
// (profiles/r08_t_*); forcing four waves on the general form cost 11 spilled registers for 2 % (profiles/r08_c_*).
// ... and likewise wherever the leaner form crosses the 128-register line (compiler's resource report, round 6: e.g. Matern-5/2 at
// KK = 2: 159 -> 128, KK = 6: 159 -> 112, KK = 7: 163 -> 118, KK = 8: 154 -> 124; RBF at KK = 7..10: 139-145 -> 117-126); pairs where it
// does not (KK = 1, 4, 5; Matern at KK = 3, 9, 10: config 4's kernel stays at 159 registers, three waves) keep the one general form.
#ifndef BASQ_BS_NOWX_FOR
#define BASQ_BS_NOWX_FOR(KK, FAM)                                                                                              \
    (((KK) == 3 && (FAM) == BASQ_FAMILY_RBF) || ((KK) == 2 && (FAM) == BASQ_FAMILY_MATERN52) ||                               \
     ((KK) == 6 && (FAM) != BASQ_FAMILY_RBF) || (KK) == 7 || (KK) == 8 || ((KK) == 9 && (FAM) != BASQ_FAMILY_MATERN52) ||      \
     ((KK) == 10 && (FAM) == BASQ_FAMILY_RBF))
#endif
#ifndef BASQ_BS_KK2_WAVES
#define BASQ_BS_KK2_WAVES 3     // three waves per SIMD asked for at KK = 2, RBF (d = 3..6): left alone the kernel takes 170 registers, two
#endif                          // over -- 6.93 -> 6.42 ms per 1e10 pairs at d = 5 (profiles/r07_x_blocksum_variants_d5_d16.txt; A/B builds: 1)
#ifndef BASQ_BS_WAVES
#define BASQ_BS_ATTR __attribute__((amdgpu_waves_per_eu((KK == 2 && FAM == BASQ_FAMILY_RBF) ? BASQ_BS_KK2_WAVES : 1)))
#else
#define BASQ_BS_ATTR __attribute__((amdgpu_waves_per_eu(BASQ_BS_WAVES, BASQ_BS_WAVES)))
#endif
template <int KK, int FAM, int JT, int XS, bool NOWX = false>
__global__ void __launch_bounds__(256) BASQ_BS_ATTR blocksum_kernel(const BlocksumArgs A_in) {
    constexpr int KP = KK * 4;
    BlocksumArgs A = A_in;
    if constexpr (NOWX) A.wx = nullptr;                          // (what the launcher checked: a compile-time fact from here on)
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
    // (scaled-argument form: the products come out as the exp's table argument, resp. as the Matern kernels' squared distance)
    constexpr double a_scale = !BASQ_BS_SCALED_ARG ? 1.0 : (FAM == BASQ_FAMILY_RBF) ? ExpScale<XS>::value : -2.0;
    if (BASQ_BS_SCALED_ARG)
        expk_init_scaled<XS>(ek, (FAM == BASQ_FAMILY_MATERN52) ? -0x1.1e3779b97f4a8p+1 * ExpScale<XS>::value
                                                               : -0x1.bb67ae8584caap+0 * ExpScale<XS>::value);
    else
        expk_init<XS>(ek);

    double a[JT][KK];
#pragma unroll
    for (int jt = 0; jt < JT; ++jt)
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) a[jt][kk] = A.nys[(long long)(j0 + jt * 16 + c) * KP + kk * 4 + g] * a_scale;

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
#ifndef BASQ_JT_LARGE_FROM
#define BASQ_JT_LARGE_FROM 5    // two row tiles per wave from this KK on.  Round 5: 5 instead of 6 -- at KK = 5 (d = 15..18) four tiles cost
#endif                          // 186-254 registers = two waves per SIMD; with two tiles 128-160 = three or four: 10.46 -> 9.92 ms per 1e10
                                // pairs at d = 16 RBF, 14.67 -> 14.48 Matern-5/2 (profiles/r07_x_blocksum_variants_d5_d16.txt; A/B builds: 6)
#define BASQ_JT_FOR(KK) ((KK) >= BASQ_JT_LARGE_FROM ? 2 : BASQ_JT_SMALL)

template <int KK, int FAM, int XS>
static int launch_blocksum(const BlocksumArgs& A, hipStream_t st) {
    constexpr int JT = BASQ_JT_FOR(KK);
    BlocksumArgs B = A;
    B.n_jgroups = (A.m + 64 * JT - 1) / (64 * JT);         // 4 waves x 16*JT rows per block
    const long long npairs = (long long)A.n_stiles * A.n_chunks;
    const long long nblk = ((npairs + 7) / 8) * 8 * B.n_jgroups;   // (set tile, chunk) pairs padded to the 8 XCDs
    if (nblk <= 0 || nblk > 0x7fffffffLL) return BASQ_EINVAL;
    if constexpr (BASQ_BS_NOWX_FOR(KK, FAM)) {
        if (B.wx == nullptr) {                             // no per-candidate kernel weights: the leaner instantiation
            hipLaunchKernelGGL((blocksum_kernel<KK, FAM, JT, XS, true>), dim3((unsigned)nblk), dim3(256), 0, st, B);
            BASQ_CHECK_LAUNCH();
            return BASQ_OK;
        }
    }
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

// The launch arguments as the kernel uses them: the kernarg segment itself (host geometry: every field stays a scalar load --
// a mutable copy costs the two registers that take the kernel from two waves per SIMD to one), or a copy patched from the
// round descriptor (GEO).
template <bool GEO>
struct SqLaunchView {
    const BlocksumArgs& A;
    const SqArgs& Q;
    __device__ SqLaunchView(const BlocksumArgs& a, const SqArgs& q) : A(a), Q(q) {}
};
template <>
struct SqLaunchView<true> {
    BlocksumArgs A;
    SqArgs Q;
    __device__ SqLaunchView(const BlocksumArgs& a, const SqArgs& q) : A(a), Q(q) {}
};

// Two waves per SIMD where the kernel fits 256 registers without spilling -- RBF with KP <= 12 (d <= 10: BASELINE config 5's shape
// sits at exactly 256; two more registers, which the descriptor-driven variant would take, halve its occupancy and its speed) --
// and the compiler's own choice everywhere else: forced on the wider variants the same attribute makes them SPILL (112-392 bytes
// per lane at KK >= 5), where they otherwise run at one wave per SIMD with their registers intact.
#define BASQ_SQ_ATTR __attribute__((amdgpu_waves_per_eu((KK <= 3 && FAM == BASQ_FAMILY_RBF) ? 2 : 1)))
template <int KK, int FAM, int JT, bool GEO>
__global__ void __launch_bounds__(256) BASQ_SQ_ATTR blocksum_sq_kernel(const BlocksumArgs A_in, const SqArgs Q_in) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    constexpr int KP = KK * 4;
    SqLaunchView<GEO> view(A_in, Q_in);
    if constexpr (GEO) view.Q.kobs += blocksum_apply_geo<KP>(view.A);   // descriptor-driven round: range (and the ko columns) from HBM
    const BlocksumArgs& A = view.A;
    const SqArgs& Q = view.Q;
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
    if (B.geo) hipLaunchKernelGGL((blocksum_sq_kernel<KK, FAM, JT, true>), dim3((unsigned)nblk), dim3(256), 0, st, B, Q);
    else hipLaunchKernelGGL((blocksum_sq_kernel<KK, FAM, JT, false>), dim3((unsigned)nblk), dim3(256), 0, st, B, Q);
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
// geo != NULL (descriptor-driven round): this rank's shard and the round's n_full come from the descriptor.
template <int FAM>
__global__ void cov_diag_kernel(const double* __restrict__ nys, int kp, int m, const double* __restrict__ cand,
                                long long Rl, long long off, long long n_full, int S, const double* __restrict__ bmatT,
                                long long ldb, const double* __restrict__ kobs, long long ldk, int n_obs,
                                double outputscale, double noise, double* __restrict__ out,
                                const long long* __restrict__ geo) {
    if (geo) {
        n_full = geo[1];
        off = geo[6];
        Rl = geo[7];
    }
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

// The part of a round's message that WSABI-M's noise cross terms contribute (see basq_sq_noise_part_geo_f64), two launches:
//   (1) BASQ_SQNP_GROUPS work-groups: group g adds mu[p] val[p] of this rank's candidates in full blocks b = g (mod groups) to
//       ws[g][s], s = the set of p, in block order (a single work-group doing all of it is a chain of ~R / S dependent loads per
//       thread: 220 us per round at config 5m's size);
//   (2) one work-group: dvec[s] = sum_g ws[g][s] (index order), dt[k] = mu val of remainder point k (k < m), then
//       part[1 + r][s] = U[r][s] dvec[s] (s < min(m, S))  + [s == S-1] sum_k U[r][k0 + k] dt[k]  (+ SOBER: U[r][s] dt[s - k0] on [k0, k1))
#define BASQ_SQNP_GROUPS 64
__global__ void __launch_bounds__(256) sq_noise_dvec_geo_kernel(const double* __restrict__ mu, const double* __restrict__ val,
                                                                const long long* __restrict__ geo, int S, double* __restrict__ ws) {
    const long long n_full = geo[1], off = geo[6], Rl = geo[7];
    long long t0l = n_full - off;                                  // first local remainder position = end of the full blocks here
    if (t0l < 0) t0l = 0;
    if (t0l > Rl) t0l = Rl;
    const long long b_lo = off / S, b_hi = (off + t0l + S - 1) / S;   // global blocks that intersect the shard's full-block part
    for (int sidx = threadIdx.x; sidx < S; sidx += blockDim.x) {
        double acc = 0.0;
        for (long long b = b_lo + blockIdx.x; b < b_hi; b += gridDim.x) {
            const long long p = b * S + sidx - off;
            if (p >= 0 && p < t0l) acc = __builtin_fma(mu[p], val[p], acc);
        }
        ws[(long long)blockIdx.x * S + sidx] = acc;
    }
}

__global__ void __launch_bounds__(1024) sq_noise_part_geo_kernel(const double* __restrict__ mu, const double* __restrict__ val,
                                                                 const long long* __restrict__ geo, const double* __restrict__ U,
                                                                 long long ldu, int q, int m, int S, int rows, int sober,
                                                                 const double* __restrict__ ws, int groups,
                                                                 double* __restrict__ part) {
    __shared__ double dvec[1024], dt[1024];
    const int tid = threadIdx.x;
    const long long n_full = geo[1], off = geo[6], Rl = geo[7];
    long long t0l = n_full - off;
    if (t0l < 0) t0l = 0;
    if (t0l > Rl) t0l = Rl;
    const long long k0 = off + t0l - n_full;
    long long k1 = k0 + (Rl - t0l);
    if (k1 > m) k1 = m;
    const int nrem = (k1 > k0) ? (int)(k1 - k0) : 0;
    if (tid < S) {
        double acc = 0.0;
        for (int g = 0; g < groups; ++g) acc += ws[(long long)g * S + tid];
        dvec[tid] = acc;
    }
    if (tid < nrem) dt[tid] = mu[t0l + tid] * val[t0l + tid];
    __syncthreads();
    const int nd = (m < S) ? m : S;
    for (int idx = tid; idx < rows * S; idx += 1024) {
        const int r = idx / S, sidx = idx - r * S;
        double v = 0.0;
        if (r >= 1 && r <= q) {
            const double* Ur = U + (long long)(r - 1) * ldu;
            if (sidx < nd && t0l > 0) v = Ur[sidx] * dvec[sidx];
            if (sidx == S - 1)
                for (int k = 0; k < nrem; ++k) v = __builtin_fma(Ur[k0 + k], dt[k], v);
            if (sober && sidx >= k0 && sidx < k1) v = __builtin_fma(Ur[sidx], dt[sidx - k0], v);
        }
        part[idx] = v;
    }
}

__global__ void axpb_strided_kernel(const double* __restrict__ x, long long n, long long stride, double a, double b,
                                    double* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = __builtin_fma(a, x[i * stride], b);
}

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
                       spec->outputscale, noise, out, (const long long*)nullptr)
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

int basq_cov_diag_geo_f64(const basq_kernel_spec* spec, const double* nys, int32_t m, const double* cand,
                          const int64_t* geo, int64_t R_max, int32_t S, const double* bmatT, int64_t ldb, const double* kobs,
                          int64_t ldk, int32_t n_obs, double noise, double* out, void* stream) {
    if (!spec_ok(spec) || !nys || !cand || !geo || !bmatT || !kobs || !out) return BASQ_EINVAL;
    if (m < 1 || R_max < 1 || S < 1 || n_obs < 1 || ldb < m || ldk < R_max) return BASQ_EINVAL;
    const int kp = basq_kp(spec->d);
    const dim3 grid((unsigned)((R_max + 255) / 256)), block(256);
#define BASQ_COV_DIAG(FAM)                                                                                              \
    hipLaunchKernelGGL((cov_diag_kernel<FAM>), grid, block, 0, (hipStream_t)stream, nys, kp, m, cand, 0LL, 0LL, 0LL, S,  \
                       bmatT, (long long)ldb, kobs, (long long)ldk, n_obs, spec->outputscale, noise, out,                \
                       (const long long*)geo)
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

int64_t basq_sq_noise_part_ws_doubles(int32_t S) { return (int64_t)BASQ_SQNP_GROUPS * (S > 0 ? S : 0); }

int basq_sq_noise_part_geo_f64(const double* mu, const double* val, const int64_t* geo, const double* U, int64_t ldu,
                               int32_t q, int32_t m, int32_t S, int32_t rows, int32_t sober, double* ws, double* part,
                               void* stream) {
    if (!mu || !val || !geo || !U || !ws || !part) return BASQ_EINVAL;
    if (q < 1 || m < 1 || S < 1 || S > 1024 || rows < q + 1 || ldu < m) return BASQ_EINVAL;
    hipLaunchKernelGGL(sq_noise_dvec_geo_kernel, dim3(BASQ_SQNP_GROUPS), dim3(256), 0, (hipStream_t)stream, mu, val,
                       (const long long*)geo, S, ws);
    BASQ_CHECK_LAUNCH();
    hipLaunchKernelGGL(sq_noise_part_geo_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, mu, val, (const long long*)geo,
                       U, (long long)ldu, q, m, S, rows, sober ? 1 : 0, ws, BASQ_SQNP_GROUPS, part);
    BASQ_CHECK_LAUNCH();
    return BASQ_OK;
}

int basq_blocksum_sq_geo_f64(const basq_kernel_spec* spec, const double* nys, int32_t m, const double* cand,
                             const double* mu, const int64_t* geo, int32_t geo_mode, int32_t S, int32_t n_chunks,
                             int32_t class_mod, int32_t class0, const double* bmatT, int64_t ldb, const double* kobs,
                             int64_t ldk, int32_t n_obs, double noise, double* Epart, void* stream) {
    if (!spec_ok(spec) || !nys || !cand || !mu || !geo || !bmatT || !kobs || !Epart) return BASQ_EINVAL;
    if (m < 1 || S < 1 || n_chunks < 1 || n_obs < 1 || geo_mode < 1 || geo_mode > 4) return BASQ_EINVAL;
    if (class_mod < 0 || class0 < 0 || (class_mod > 0 && class0 + n_chunks > class_mod)) return BASQ_EINVAL;
    if (class_mod > 0 && geo_mode != 1) return BASQ_EINVAL;       // residue classes cover the regular region only
    const int jt = BASQ_JT_FOR(basq_kp(spec->d) / 4);
    if (ldb < (((int64_t)m + 16 * jt - 1) / (16 * jt)) * (16 * jt) || ldk < 1) return BASQ_EINVAL;
    BlocksumArgs A;
    A.nys = nys; A.cand = cand; A.mu = mu; A.wx = nullptr; A.Xpart = Epart; A.totpart = nullptr;
    A.Rl = 0; A.off = 0; A.n_full = 0; A.blk_lo = 0; A.blk_hi = 0; A.blk_per_chunk = 1;   // set on the device
    A.m = m; A.S = S; A.n_chunks = n_chunks;
    A.class_mod = class_mod; A.class0 = class0;
    A.geo = (const long long*)geo; A.geo_mode = geo_mode;
    A.n_stiles = (S + 15) / 16;
    SqArgs Q;
    Q.bmatT = bmatT; Q.kobs = kobs; Q.ldb = ldb; Q.ldk = ldk; Q.ko = (n_obs + 3) / 4;
    Q.outputscale = spec->outputscale; Q.noise = noise;
    return dispatch_blocksum_sq(basq_kp(spec->d) / 4, spec->family, A, Q, (hipStream_t)stream);
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

int basq_blocksum_geo_f64(const basq_kernel_spec* spec, const double* nys, int32_t m, const double* cand,
                          const double* mu, const double* wx, const int64_t* geo, int32_t geo_mode, int32_t S,
                          int32_t n_chunks, int32_t class_mod, int32_t class0, double* Xpart, double* totpart,
                          void* stream) {
    if (!spec_ok(spec) || !nys || !cand || !mu || !Xpart || !totpart || !geo) return BASQ_EINVAL;
    if (m < 1 || S < 1 || n_chunks < 1 || geo_mode < 1 || geo_mode > 5) return BASQ_EINVAL;
    if (class_mod < 0 || class0 < 0 || (class_mod > 0 && class0 + n_chunks > class_mod)) return BASQ_EINVAL;
    // residue classes cover full blocks only: the regular region (mode 1), or -- mode 5 -- the fewer than class_mod full blocks
    // behind it, whose block index modulo class_mod numbers them (the regular region is a multiple of class_mod blocks)
    if (class_mod > 0 && geo_mode != 1 && geo_mode != 5) return BASQ_EINVAL;
    if (geo_mode == 5 && class_mod < 2) return BASQ_EINVAL;
    BlocksumArgs A;
    A.nys = nys; A.cand = cand; A.mu = mu; A.wx = wx; A.Xpart = Xpart; A.totpart = totpart;
    A.Rl = 0; A.off = 0; A.n_full = 0; A.blk_lo = 0; A.blk_hi = 0; A.blk_per_chunk = 1;   // set on the device
    A.m = m; A.S = S; A.n_chunks = n_chunks;
    A.class_mod = class_mod; A.class0 = class0;
    A.geo = (const long long*)geo; A.geo_mode = geo_mode;
    A.n_stiles = (S + 15) / 16;
    return dispatch_blocksum(basq_kp(spec->d) / 4, spec->family, A, (hipStream_t)stream, blocksum_exp_scheme(spec));
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

}  // extern "C"
