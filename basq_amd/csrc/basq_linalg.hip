// basq_linalg.hip -- dense linear algebra of the path on the f64 matrix cores: GEMM / tall-skinny GEMM (range finder,
// projections), message finalisation, Cholesky / triangular solves / CholeskyQR, Box-Muller; and their C-ABI entries.
#include "basq_common.hpp"

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


extern "C" {

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
