/*
 * basq_hip.h -- C ABI of libbasq_hip.so: MI355X (gfx950) kernels for the kernel-recombination
 * hot path of ma921/BASQ (BASQ/_rchq.py), float64.
 *
 * The reference has no FFI: its boundary is the Python function
 *     recombination(pts_rec, pts_nys, num_pts, kernel, device, init_weights)   BASQ/_rchq.py:4-25
 * reached through BASQ.run_rchq (BASQ/_basq.py:59-80) and KernelQuadrature.rchq
 * (BASQ/_quadrature.py:29-51).  This library is what a binding of that path would call; the Python
 * shim basq_amd/_rchq.py (ctypes, tensor.data_ptr(), current HIP stream) keeps the reference's
 * signatures.  Each entry point below names the reference lines it replaces.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless the name ends in _host; the caller owns all buffers;
 *    the library allocates nothing and keeps no global mutable state;
 *  - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises;
 *  - return value: 0 on success, a negative BASQ_E* code otherwise; never throws, never exits;
 *  - matrices are row-major, float64; indices are int64 unless stated.
 *
 * Packed operands.  Points are pre-packed once per batch into rows of KP = basq_kp(d) doubles so that
 * the pairwise exponent argument is ONE dot product on the f64 matrix cores:
 *     role A (left / Nystrom side):    [ (x-c)/l ... , 0.. , h, 1 ]     h = -1/2 |(x-c)/l|^2
 *     role B (right / candidate side): [ (y-c)/l ... , 0.. , 1, h ]     (h and 1 in the LAST two slots)
 *     A_row . B_row = -1/2 |(x-y)/l|^2   (c = centring vector, l = lengthscale)
 * (the centring mirrors gpytorch's mean-centred squared distance, see oracle/kernels_oracle.py).
 */
#ifndef BASQ_HIP_H
#define BASQ_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BASQ_ABI_VERSION 15

/* error codes */
#define BASQ_OK            0
#define BASQ_EINVAL       -1   /* bad argument (null pointer, negative size, unsupported d, ...) */
#define BASQ_ELAUNCH      -2   /* HIP launch error (see hipGetLastError on the caller side)       */
#define BASQ_EUNSUPPORTED -3   /* kernel family / size not compiled in                            */

/* kernel families: the stationary kernels the reference selects at BASQ/_parameters.py:192-208 */
#define BASQ_FAMILY_RBF      0   /* s2 * exp(-r^2/2)                           */
#define BASQ_FAMILY_MATERN52 1   /* s2 * (1 + sqrt5 r + 5 r^2/3) exp(-sqrt5 r) */
#define BASQ_FAMILY_MATERN32 2   /* s2 * (1 + sqrt3 r) exp(-sqrt3 r)           */

#define BASQ_ROLE_A 0
#define BASQ_ROLE_B 1

#define BASQ_MAX_DIM 38   /* KP <= 40 */

typedef struct basq_kernel_spec {
    int32_t family;        /* BASQ_FAMILY_*                                        */
    int32_t d;             /* input dimension, 1..BASQ_MAX_DIM                     */
    double  lengthscale;   /* single shared lengthscale (no ARD), > 0              */
    double  outputscale;   /* ScaleKernel outputscale s2                           */
    int32_t flags;         /* BASQ_SPEC_* bits                                     */
    int32_t reserved;      /* 0                                                    */
} basq_kernel_spec;

/* flags: the fused block sums (basq_blocksum_f64 / _geo) evaluate exp() with the 2048-entry table + cubic (relative error
 * 1e-17) instead of the 4096-entry table + quadratic (2.5e-14, one fp64 instruction less per kernel value).  Set for GP
 * posteriors (BASQ/_gp.py:259-277): there the message is k - k(.,X) W k(X,.), a cancellation that amplifies kernel-value
 * errors by up to the conditioning of the observation Gram. */
#define BASQ_SPEC_ACCURATE_EXP 1

const char* basq_strerror(int code);
int         basq_abi_version(void);

/* Measurement aid, no counterpart in the reference: the shader clock in MHz over each of the next n periods of period_us
 * microseconds -> out[n] (device).  bench.py launches it on a second stream beside the dominant kernel: the clock a launch
 * runs at inside a batch is what its roofline is set by.  n * period_us <= 1e6. */
int basq_shader_clock_mhz(double* out, int n, int period_us, void* stream);

/* KP: packed row length (multiple of 4, >= d + 2). */
int basq_kp(int d);

/* Column means of X[n,d] -> mean[d]  (the centring vector; gpytorch centres on the first operand). */
int basq_col_mean_f64(const double* X, int64_t n, int d, double* mean, void* stream);

/* Pack X[n,d] into out[n,KP] for `role`; `center` may be NULL (no centring). */
int basq_pack_points_f64(const basq_kernel_spec* spec, const double* X, int64_t n, const double* center,
                         int role, double* out, void* stream);

/*
 * Dense kernel matrix K[na,nb] = k(A_i, B_j) from packed operands (ldk = row stride of K, >= nb).
 * Replaces the `kernel(x, y)` callable of the reference where a dense block is really needed:
 * the Nystrom Gram `kernel(pt, pt)` (BASQ/_rchq.py:29) and K(X,X) of the quadrature (BASQ/_quadrature.py:62).
 */
int basq_gram_f64(const basq_kernel_spec* spec, const double* packA, int64_t na, const double* packB, int64_t nb,
                  double* K, int64_t ldk, void* stream);

/*
 * Kernel mat-vec: out[i] = bias + sum_j k(A_i, B_j) * v[j], i < na   (GP posterior mean of predict(),
 * BASQ/_gp.py:213-230, with v = mean_cache, bias = constant mean).
 */
int basq_kernel_matvec_f64(const basq_kernel_spec* spec, const double* packA, int64_t na, const double* packB,
                           int64_t nb, const double* v, double bias, double* out, void* stream);

/*
 * Fused block sums -- the hot loop BASQ/_rchq.py:79-86 (+ tot_weights :90 and the ragged tail :91-99).
 *
 * The rank holds `Rl` surviving candidates (rows of `cand`, role-B packed, in ascending global
 * position) starting at global position `off`; global position p belongs to set  p % S  when
 * p < n_full (= number_of_el * S) and to set S-1 otherwise (the tail).  For every Nystrom row j < m
 * (role-A packed `nys`, allocated with rows padded to a multiple of 64) and set s < S:
 *     Xpart[c][j][s]  = sum over this rank's candidates p of chunk c in set s of  k(nys_j, cand_p) * mu_p * (wx ? wx_p : 1)
 *     totpart[c][s]   = sum of mu_p over the same candidates
 * The candidate blocks are split into `n_chunks` contiguous chunks (more parallelism; the consumer
 * adds the chunks in index order, so results do not depend on scheduling).  Chunk n_chunks-1 also
 * receives the tail.  Xpart: [n_chunks, m, S], totpart: [n_chunks, S]; both fully overwritten.
 * The output is NOT multiplied by outputscale (basq_project_f64 applies it).
 *
 * Residue classes (class_mod > 0): chunk c sums the blocks b with  b % class_mod == class0 + c  instead of a contiguous
 * range (class0 + n_chunks <= class_mod; the rank's range must hold full blocks only, off + Rl <= n_full).  Summing
 * all class_mod classes gives the same block sums; keeping them apart makes the NEXT rounds free, see
 * basq_regroup_classes_f64.
 */
int basq_blocksum_f64(const basq_kernel_spec* spec, const double* nys, int32_t m, const double* cand,
                      const double* mu, const double* wx, int64_t Rl, int64_t off, int64_t n_full, int32_t S,
                      int32_t n_chunks, int32_t class_mod, int32_t class0, double* Xpart, double* totpart, void* stream);

/*
 * The same contraction per chunk (Ut [m, q] as above) -- out[c] = [ totpart[c] ; outputscale * U @ Xpart[c] ], each (q+1) x S --
 * for residue-class chunks: the classes' MESSAGES (100 x 200 doubles each) are all the later rounds of an epoch need
 * (basq_regroup_classes_f64 applies to them row by row), so the [m, S] class partials can be dropped right after this
 * call.  work: n_chunks * ksplit * q * S doubles (the library picks nz <= ksplit K slices per chunk so that the launch fills
 * one round of the chip's wave slots; for q <= 208 the chunks run as one batched launch of the tall-skinny kernel of
 * basq_skinny_gemm_f64, Xpart[c]^T [S, m] @ Ut [m, q], every partial streamed from HBM once).
 * basq_sum_parts_f64: out[e] = sum_p parts[p][e] in index order (the round's message = the sum of its class messages).
 */
int basq_project_chunks_f64(const double* Ut, int32_t q, int32_t m, const double* Xpart, const double* totpart,
                            int32_t n_chunks, int32_t S, double outputscale, int32_t ksplit, double* work, double* out,
                            void* stream);
int basq_sum_parts_f64(const double* parts, int32_t n_parts, int64_t n, double* out, void* stream);

/*
 * Block sums of the next round WITHOUT evaluating the kernel again.  When a round keeps exactly n_keep = S/2 sets
 * (BASQ/_rchq.py:107-130), the survivor at (block b, kept set of rank k) moves to position b*S/2 + k: block b/2, set
 * (b % 2) * S/2 + k, and its weight is rescaled by w_star[k] / tot[kept[k]] (:113-114).  So with the sums of this round
 * held per residue class of the block index,
 *     Tin[c][j][s] = sum over blocks b == c (mod C) of k(nys_j, x_{b,s}) mu_{b,s},
 * the next round's are a gather and a rescale:
 *     Tout[c'][j][par * S/2 + k] = (Tin[2c' + par][j][kept[k]] * w_star[k]) / tot[kept[k]],      c' < C/2, par in {0, 1}
 * -- half as many classes, so C classes pay for log2(C) rounds.  rows = rows of each [rows, S] class matrix: the map is
 * linear, so it applies equally to the class MESSAGES [ tot ; U @ T ] of basq_project_chunks_f64 (rows = q + 1), which
 * is how the engine uses it.  Valid for the blocks the classes cover; the caller evaluates the few remaining candidates
 * (blocks beyond a multiple of C, the ragged tail) directly.  C even, S even.
 */
int basq_regroup_classes_f64(const double* Tin, int32_t rows, int32_t S, int32_t C, const int32_t* kept,
                             const double* w_star, const double* tot, double* Tout, void* stream);

/*
 * Nystrom-feature contraction BASQ/_rchq.py:88-90:  out[0][s] = sum_c totpart[c][s];
 *     out[1+r][s] = outputscale * sum_j U[r][j] * (sum_c Xpart[c][j][s])        r < q
 * on the f64 matrix cores (v_mfma_f64_16x16x4_f64): the chunk partials are first added in chunk order
 * (one streaming pass), then ONE GEMM split `ksplit` ways along K whose slabs are re-added in index order.
 * `work` holds (n_chunks > 1 ? m*S : 0) + ksplit*q*S doubles, 16-byte aligned; m*S must be even if n_chunks > 1.  `out` is [(q+1), S]: the per-rank message
 * of the multi-GPU all-gather; it is NOT yet divided by the set weights.
 * The basis is passed TRANSPOSED: Ut [m, q] row-major (Ut[j][r] = U[r][j]) -- the MFMA row groups then read whole
 * 128-byte lines of it (a 10x faster contraction than from the [q, m] layout; the caller transposes once per batch).
 */
int basq_project_f64(const double* Ut, int32_t q, int32_t m, const double* Xpart, const double* totpart,
                     int32_t n_chunks, int32_t S, double outputscale, int32_t ksplit, double* work, double* out,
                     void* stream);

/*
 * Sum `n_parts` rank messages [msg_rows, S] in index order, divide feature rows 1..q by row 0 (block
 * barycentres, BASQ/_rchq.py:101) and write the matrix the reduction decomposes
 * (BASQ/_rchq.py:138-140):  XcarT[0][s] = 1, XcarT[1+r][s] = feature_r(s) / tot[s];  tot[s] -> tot_out.
 * Optional additive term (predictive_covariance's diagonal noise, BASQ/_gp.py:275-276, which the
 * reference adds to entry [k][k] of EVERY kernel block it builds): if diagU != NULL,
 *     feature_r(s) += diag_noise * weight(s) * diagU[r*ld_diag + s]                  for s < n_diag
 * with weight = the weight of set s summed over the FULL blocks: message row `diag_wrow` (0 = the set weights;
 * q+1 = an extra message row, used by the WSABI-L kernel whose block sums carry a per-candidate factor), minus --
 * for the last set -- the tail weights.  The ragged tail is its own block (BASQ/_rchq.py:91-99), whose entry
 * [k][k] pairs tail point k with Nystrom row k and lands in the LAST set: with `diag_tail_row` != 0 naming a
 * message row that holds the tail points' weights (tail[k], zero beyond the tail),
 *     feature_r(S-1) += diag_noise * sum_{k < n_tail_diag} tail[k] * diagU[r*ld_diag + k].
 * msg_rows >= q + 1;  n_diag, n_tail_diag <= ld_diag.
 */
int basq_finalize_f64(const double* parts, int32_t n_parts, int32_t msg_rows, int32_t q, int32_t S,
                      const double* diagU, int64_t ld_diag, int32_t n_diag, double diag_noise, int32_t diag_wrow,
                      int32_t diag_tail_row, int32_t n_tail_diag, double* XcarT, double* tot_out, void* stream);

/*
 * Null-space basis of the wide [s, M] matrix XcarT, replacing the full SVD of BASQ/_rchq.py:140-143
 * (`u, _, _ = torch.linalg.svd(X.T)`, `Phi = u[:, -(M-s):]`).  The elimination's pivots depend on the basis
 * LAPACK returns, not only on the null space; that basis is rows s..M-1 of P^T, P = G_0 ... G_{s-1} the product
 * of the right Householder reflectors of gesdd's bidiagonal reduction (dgebrd, m < n, dlarfg sign convention) --
 * the rotations that follow never touch those rows.  This entry generates the same reflectors on the GPU
 * (matrix resident in registers: one work-group for M <= 256, a cluster of four exchanging one message per step
 * through `ws` for M <= 512, see basq_reduction_ws_doubles) and applies them to the unit vectors e_s..e_{M-1}:
 *     PhiT [M-s, M]  (rows = null vectors, what basq_car_eliminate_f64 consumes).
 * Scratch: V [s, M] (reflector vectors; also row storage when s*M exceeds the LDS), tau [s].
 * Agreement with the host LAPACK rows is at rounding level (~1e-13); 1 <= s < M <= 1024.
 * info (optional, device int32[1]): 0, or 2 = a cluster work-group gave up waiting for its siblings (bounded spins:
 * the eight work-groups of a cluster must be co-resident, which a GPU shared with other work may not grant within the
 * limit); PhiT is then poisoned with NaNs.  Passing ws = NULL selects the single-work-group kernels for every shape
 * (slower for M > 256, no co-residency requirement): the caller's retry path.
 */
int basq_nullspace_f64(const double* XcarT, int32_t s, int32_t M, double* V, double* tau, double* PhiT, double* ws,
                       int32_t* info, void* stream);

/*
 * Workspace (in doubles) that basq_nullspace_f64 / basq_car_eliminate_f64 need in `ws` for an [s, M] reduction:
 * 0 when the shape runs on one compute unit (M <= 256: ws may be NULL), otherwise the larger of the elimination's ring of
 * pivot rows (256 < M <= 448: 16 + 2 (M - s)(64 ceil(M / 64) + 4) doubles, 1.4 MB at 200 x 400) and the message ring of the
 * 8-work-group cluster kernels (M = 2n = 400 at n = 200: the 200 x 400 matrix does not fit one CU's registers): tagged
 * 16-byte granules {tag, low word, tag, high word} -- the data is its own flag.  The caller owns the buffer (the library
 * allocates nothing); the entries zero its words themselves before every launch (tags count the steps of ONE launch).  Without a
 * workspace (ws == NULL) such shapes fall back to slower single-work-group kernels.
 */
int64_t basq_reduction_ws_doubles(int32_t s, int32_t M);

/*
 * Caratheodory elimination -- the loop of Tchernychova_Lyons_CAR, BASQ/_rchq.py:146-175, in the
 * reference's floating-point op order (separate multiply / subtract, outer product then divide).
 * PhiT is the null-space basis as ROWS: PhiT[k][i] = Phi[i][k], [M-s, M] (the last M-s rows of the
 * full Vh of the SVD at :140-143); it is destroyed.  mu [M] holds the set weights and is READ ONLY (since ABI 13: the
 * in-place update of earlier versions cost every caller a copy per round -- the reduced weights are w_star).  Outputs: keep_rank[M] (rank among survivors
 * or -1), kept[<=s] ascending survivor ids, w_star[<=s], info[0] = n_keep, info[1] = status
 * (0 ok, 1 = a null vector had no positive entry: the reference would raise at :152; 2 = a cluster kernel's
 * bounded spin timed out -- never in a healthy run).  Null vectors live in registers: for M <= 256 and M - s <= 112 one
 * work-group whose 16 waves own CONSECUTIVE null vectors (seven each) -- a wave consumes the pivots published before its block
 * (one rank-1 update of its rows per pivot) and then runs the ratio tests of its own block without leaving the wave; every pivot
 * row is written to LDS once (car_eliminate_ring_kernel, round 4: 122 us against 174 at 100 x 200); for 256 < M <= 448 the same
 * block scheme over several work-groups of 8 waves (4 rows per wave), the pivot rows as tagged granules in `ws`, one slot per
 * pivot, zeroed by this entry before the launch (car_eliminate_gring_kernel, round 4: 307 us against 520 at 200 x 400; a wave
 * only ever waits for EARLIER blocks, so the work-groups need not be co-resident; BASQ_CAR_GRING=0: the cluster kernel); an
 * 8-work-group cluster exchanging one message per step through `ws` for the remaining M <= 512 (see basq_reduction_ws_doubles;
 * ws may be NULL otherwise); M <= 1024.  The divisions of the ratio
 * test are the IEEE expansion without its scaling steps (operands and quotients far from the ends of the exponent range -- what
 * the Markstein quotient of the rank-1 update has always assumed); BASQ_CAR_RING=0 in the environment selects the LDS-resident
 * kernel of rounds 1-3 (A/B).
 */
int basq_car_eliminate_f64(double* PhiT, double* mu, int32_t M, int32_t s, int32_t* keep_rank, int32_t* kept,
                           double* w_star, int32_t* info, double* ws, void* stream);

/*
 * Survivor re-weighting and order-preserving compaction, BASQ/_rchq.py:107-130.
 * For each local candidate p (global position off+p): its set's survivor rank kr = keep_rank[set];
 * dropped if kr < 0; otherwise mu' = (mu * w_star[kr]) / tot[set] (multiply, then divide: :113-114)
 * and the row moves to new local position
 *     (blk * n_keep + kr) - new_off              for block positions (blk = position / S)
 *     (nb * n_keep + position - n_full) - new_off for tail positions
 * cand/mu/gid/wx -> cand_out/mu_out/gid_out/wx_out (wx may be NULL).  new_off is computed by the
 * host from the same closed form (basq_amd/_partition.py).
 */
int basq_reweight_compact_f64(const double* cand, const double* mu, const int64_t* gid, const double* wx,
                              int64_t Rl, int64_t off, int64_t n_full, int32_t S, int32_t kp,
                              const int32_t* keep_rank, const double* w_star, const double* tot, int32_t n_keep,
                              int64_t new_off, double* cand_out, double* mu_out, int64_t* gid_out, double* wx_out,
                              void* stream);

/*
 * Device-resident round descriptor: the divide-and-conquer loop (BASQ/_rchq.py:76-130) without a host round trip per
 * round.  The number of survivors of a round depends on the data only through two facts -- how many sets the elimination
 * kept and whether the last set (which owns the ragged tail, :91-99) is among them -- so the next round's geometry is a
 * closed form of the previous one (basq_amd/_partition.py) that a one-thread kernel can evaluate:
 *   geo[8] = { R, n_full = (R / S) * S, reg_hi, violation, nb, n_tail, off, Rl }     (int64)
 *       R .. n_tail describe the round GLOBALLY; [off, off + Rl) is the slice of the live positions THIS rank holds (one
 *       rank: [0, R)), so the same entries drive the pool-sharded multi-GPU rounds without a host wait.
 *   basq_round_next_i64: R' = nb * n_keep + (n_tail if keep_rank[S-1] >= 0), n_keep / status from the elimination's
 *       info word; off' / Rl' = the survivors before position off / inside the shard (closed form of
 *       basq_amd/_partition.py: b * n_keep + #{kept sets < p mod S} below n_full, the tail behind it if set S-1 survived).  class_mode > 0: the next round evaluates its block sums afresh with that many residue classes
 *       (reg_hi' = S * the largest multiple of class_mode blocks), -1: it inherits regrouped classes (the regular
 *       region halves), 0: no classes.  violation is sticky: status != 0, or expect_half != 0 and 2 n_keep != S (the host
 *       had already enqueued a regrouping that needs exactly half of the sets): the host then repeats the batch with
 *       one read-back per round.  After a violation the descriptor describes an EMPTY round (R' = 0), so every launch
 *       already enqueued for later rounds is a no-op inside the buffers the host sized for the expected counts.
 *   basq_blocksum_geo_f64: basq_blocksum_f64 with the candidate range taken from the descriptor -- geo_mode 1: positions
 *       [0, reg_hi) (class_mod > 0 allowed), 2: [reg_hi, R), 3: [0, R), each intersected with this rank's shard (the
 *       pointers, which address the shard's first candidate, are advanced on the device); 4: the ragged remainder
 *       [n_full, R) as ONE block of its own -- remainder point k in set k -- the first of the two counts SOBER/_rchq.py
 *       gives the remainder (:127-135; the caller drops the set weights of that launch: the reference adds none there);
 *       5 (ABI 15; class_mod >= 2 required): the full blocks BEHIND the regular region, [reg_hi, n_full), one chunk per block --
 *       the regular region is a multiple of class_mod blocks, so chunk c is block reg_hi / S + c (fewer than class_mod of them;
 *       a chunk without a block is written as zeros).
 *   basq_reweight_compact_geo_f64: basq_reweight_compact_f64 with off, Rl, n_full from the descriptor, new_off from the
 *       NEXT round's descriptor (geo_next, written by basq_round_next_i64 just before) and n_keep from info[0]; the
 *       launch is sized for R_max >= Rl candidates and the outputs hold out_rows
 *       rows.  Nothing is written when the round violates what the host assumed when it sized them: geo[3] != 0 (an earlier
 *       round's flag), info[1] != 0 (the elimination stopped early or timed out) or expect_keep >= 0 and
 *       n_keep != expect_keep; no destination row >= out_rows is ever written.
 * Launch grids never depend on R; buffers are sized from host-side upper bounds.
 */
/* Descriptor-driven siblings of the noise-diagonal bookkeeping (BASQ/_gp.py:275-276 on the ragged tail block):
 *   basq_finalize_geo_f64: basq_finalize_f64 with the tail length taken from geo[5]; n_tail_diag is then only a cap.
 *   basq_tail_weights_geo_f64: out[k] = mu * wx (wx may be NULL) of tail point k (global position n_full + k) where this rank
 *       holds it, 0 elsewhere and for n_tail <= k < S: the message row that carries the tail weights (summed over the
 *       ranks with the rest of the message). */
int basq_finalize_geo_f64(const double* parts, int32_t n_parts, int32_t msg_rows, int32_t q, int32_t S,
                          const double* diagU, int64_t ld_diag, int32_t n_diag, double diag_noise, int32_t diag_wrow,
                          int32_t diag_tail_row, int32_t n_tail_diag, const int64_t* geo, double* XcarT, double* tot_out,
                          void* stream);
int basq_tail_weights_geo_f64(const double* mu, const double* wx, const int64_t* geo, int32_t S, double* out, void* stream);
int basq_round_next_i64(const int64_t* geo, const int32_t* info, const int32_t* keep_rank, int32_t S, int32_t class_mode,
                        int32_t expect_half, int64_t* geo_next, void* stream);
/* basq_regroup_classes_f64 + basq_round_next_i64 in ONE launch (the two that follow every elimination inside an epoch; both read
 * the elimination's outcome, neither reads the other's): arguments as for the two entries. */
int basq_regroup_round_next_f64(const double* Tin, int32_t rows, int32_t S, int32_t C, const int32_t* kept,
                                const double* w_star, const double* tot, double* Tout, const int64_t* geo,
                                const int32_t* info, const int32_t* keep_rank, int32_t class_mode, int32_t expect_half,
                                int64_t* geo_next, void* stream);
int basq_blocksum_geo_f64(const basq_kernel_spec* spec, const double* nys, int32_t m, const double* cand,
                          const double* mu, const double* wx, const int64_t* geo, int32_t geo_mode, int32_t S,
                          int32_t n_chunks, int32_t class_mod, int32_t class0, double* Xpart, double* totpart,
                          void* stream);

/*
 * Epochs without a pairwise evaluation inside (ABI 15; BASQ/_rchq.py:76-130 over the rounds of an epoch).  The candidates the
 * residue classes do not cover -- e < C full blocks behind the regular region and t < S tail points (:91-99) -- are carried as
 * MESSAGE COLUMNS: a round's buffer is  parts = [C class messages | 1 fold slot | E block slots | 1 tail slot], each
 * [rows, S] (row 0 = set weights, rows 1.. = projected block sums; the fold slot = the sum the finalize kernel adds behind the
 * classes: blocks in index order, then the tail's columns into set S - 1); block slot b < e holds one column per point of that block
 * (its set = the column index), the tail slot one column per tail point.  A round keeps n_keep = S / 2 sets and moves the
 * survivor of (block b, kept rank k) to position b n_keep + k, the tail -- if set S - 1 is kept -- behind; the regular
 * survivors fill the next regular region exactly, so next round's columns are a gather + rescale ((v * w_star[k]) / tot[s],
 * the order of :113-114) of this round's.
 *   (In an epoch's FIRST round the fold slot is the projected block sum of the ordinary irregular chunk -- basq_blocksum_geo_f64
 *   mode 2 -- and the columns are basq_blocksum_geo_f64 mode 5 (one chunk per block) and mode 4 (the tail, point k in set k),
 *   projected like any other chunk.)
 *   basq_epoch_turn_f64: ONE launch behind an elimination inside an epoch: Pout = next round's buffer [C / 2 | fold | E_out
 *       | tail] from Pin [C | fold | E_in | tail] and the elimination's outcome (kept, keep_rank, w_star, tot, info), and
 *       geo_next = basq_round_next_i64(geo, ..., class_mode -1, expect_half 1).  A round that did not keep exactly half of
 *       the sets (or failed, or carries the sticky flag) leaves zeros; e > E_in or e' > E_out likewise (never by the bounds
 *       the host derives: e' <= (E_in S / 2 + S - 1) / S).
 *   basq_reweight_compact_rounds_f64: basq_reweight_compact_geo_f64 for n_rounds <= 8 consecutive rounds at once -- geo points
 *       at the first round's descriptor row (rows of 8, consecutive), the outcome arrays are HOST arrays of n_rounds device
 *       pointers; the rank's shard [off, off + Rl) of the first round and its new offset come from the descriptors (geo[6], geo[7]; row
 *       n_rounds = the descriptor behind the last of the rounds); per round mu <- (mu * w_star[k]) / tot[set], in round order.
 */
int basq_epoch_turn_f64(const double* Pin, int32_t C, int32_t E_in, double* Pout, int32_t E_out, int32_t rows, int32_t S,
                        const int32_t* kept, const int32_t* keep_rank, const double* w_star, const double* tot,
                        const int32_t* info, const int64_t* geo, int64_t* geo_next, void* stream);
int basq_reweight_compact_rounds_f64(const double* cand, const double* mu, const int64_t* gid, const double* wx,
                                     const int64_t* geo, int32_t n_rounds, const int32_t* const* keep_rank,
                                     const double* const* w_star, const double* const* tot, const int32_t* const* info,
                                     int64_t R_max, int32_t S, int32_t kp, int64_t out_rows, int32_t expect_keep,
                                     double* cand_out, double* mu_out, int64_t* gid_out, double* wx_out, void* stream);

int basq_reweight_compact_geo_f64(const double* cand, const double* mu, const int64_t* gid, const double* wx,
                                  const int64_t* geo, const int64_t* geo_next, const int32_t* info, int64_t R_max,
                                  int32_t S, int32_t kp,
                                  const int32_t* keep_rank, const double* w_star, const double* tot, int64_t out_rows,
                                  int32_t expect_keep, double* cand_out, double* mu_out, int64_t* gid_out, double* wx_out,
                                  void* stream);

/* Initial state: mu[p] = 1/N_total (BASQ/_rchq.py:53), gid[p] = gid0 + p (:55). */
int basq_init_state_f64(double* mu, int64_t* gid, int64_t Rl, int64_t gid0, int64_t n_total, void* stream);

/*
 * Block sums of a dense per-pair matrix the CALLER evaluated (two users):
 *   square == 0: the hot loop BASQ/_rchq.py:79-99 for an opaque `kernel` callable (the reference accepts ANY Python
 *                callable (X[a,d], Y[b,d]) -> Tensor[a,b], BASQ/_rchq.py:8,16; tutorial 02 "arbitrary kernel"):
 *                C = kernel(pts_nys, chunk) is produced by the callable on the device, chunk by chunk, and
 *                    E[j][s] += scale * sum over the nc candidates p of this chunk with set(p) = s of  mu[p] * C[j][p]
 *   square != 0: WSABI-M's extra term (BASQ/_wsabi.py:227-249: CLy = mu_x cov mu_y + 0.5 cov^2; the first product is
 *                linear in the block sums and runs through basq_blocksum_f64, the square is not):
 *                    E[j][s] += scale * sum ... mu[p] * C[j][p]^2
 * C [m, nc] (row stride ldc) = values of the Nystrom rows against nc consecutive candidates whose first GLOBAL
 * position is pg0 (set(p) = p % S below n_full, S-1 from n_full on); mu is indexed from the chunk start.
 * E [m, S] is accumulated into (zero it first); chunks must be submitted in position order (fixed summation order).
 * tot (may be NULL): tot[s] += sum of mu[p] over the same candidates -- the set weights `tot_weights` (BASQ/_rchq.py:90,
 * :98), unscaled -- in the same launch.
 */
int basq_dense_blocksum_f64(const double* C, int32_t m, int64_t nc, int64_t ldc, const double* mu, int64_t pg0,
                            int64_t n_full, int32_t S, double scale, int32_t square, double* E, double* tot, void* stream);

/*
 * WSABI-M's non-linear term, fused (BASQ/_wsabi.py:227-249: CLy = mu_x cov mu_y + 0.5 cov^2, cov = the GP's predictive
 * covariance BASQ/_gp.py:233-277): block sums of the SQUARED covariances straight from the packed points,
 *     Epart[c][j][s] = sum over the candidates p of chunk c with set(p) = s of  (mu[p] / 2) * cov(nys_j, y_p)^2,
 *     cov(nys_j, y_p) = outputscale * k(nys_j, y_p) - sum_o bmatT[o][j] * kobs[o][p]  (+ noise where the pair is entry
 *     [kappa][kappa] of its kernel block: kappa = p % S below n_full, p - n_full in the ragged tail, _gp.py:275-276).
 * bmatT [n_obs4, ldb] = (k(nys, Xobs) W)^T, kobs [n_obs4, ldk] = outputscale * k(Xobs, y_p) at LOCAL candidate positions
 * (basq_gram_f64 of the packed observations against `cand`); n_obs4 = n_obs rounded up to 4, the extra rows zero;
 * ldb >= m rounded up to 64, the extra columns zero; ldk >= Rl.  nys / cand / mu / Rl / off / n_full / S / n_chunks as
 * for basq_blocksum_f64 (contiguous chunks).  No [m, candidates] matrix is formed; replaces the dense
 * covariance chunks + library GEMM + basq_dense_blocksum_f64(square = 1) of the unfused path.
 * class_mod / class0: residue classes of the block index as for basq_blocksum_f64 (the squared term is a per-pair block sum
 * like any other: with noise = 0 its class sums regroup over the rounds of an epoch, basq_regroup_classes_f64; the noise
 * cross terms, which sit on ONE Nystrom row per candidate, then come from basq_cov_diag_f64).
 */
int basq_blocksum_sq_f64(const basq_kernel_spec* spec, const double* nys, int32_t m, const double* cand,
                         const double* mu, int64_t Rl, int64_t off, int64_t n_full, int32_t S, int32_t n_chunks,
                         int32_t class_mod, int32_t class0,
                         const double* bmatT, int64_t ldb, const double* kobs, int64_t ldk, int32_t n_obs, double noise,
                         double* Epart, void* stream);

/*
 * The likelihood noise INSIDE WSABI-M's squared covariance (BASQ/_gp.py:275-276 under BASQ/_wsabi.py:240-242), per candidate:
 * predictive_covariance adds the noise to entry [kappa][kappa] of every kernel block, and candidate p meets it on ONE Nystrom
 * row, kappa = p % S below n_full, p - n_full in the ragged remainder.  0.5 (c + noise)^2 = 0.5 c^2 + (noise c + 0.5 noise^2):
 *     out[p] = noise * cov(nys_kappa, y_p) + 0.5 noise^2     (cov without noise, operands as for basq_blocksum_sq_f64;
 *                                                              0 where kappa >= m)
 * for the Rl local candidates (positions off ..).  The caller sums mu[p] * out[p] per set (basq_dense_blocksum_f64 on the
 * [1, Rl] row) and adds U[:, kappa] times it to the round's message.
 */
int basq_cov_diag_f64(const basq_kernel_spec* spec, const double* nys, int32_t m, const double* cand, int64_t Rl,
                      int64_t off, int64_t n_full, int32_t S, const double* bmatT, int64_t ldb, const double* kobs,
                      int64_t ldk, int32_t n_obs, double noise, double* out, void* stream);

/*
 * WSABI-M (BASQ/_wsabi.py:227-249) in the DESCRIPTOR-DRIVEN rounds (no host wait per round; ABI 14): the three entries above
 * with the candidate range read on the device from a round descriptor (geo: see basq_round_next_i64 -- {R, n_full, reg_hi,
 * violation, nb, n_tail, shard offset, shard length}), so that the host can enqueue a round of a WSABI-M batch before it knows
 * how many candidates survived the previous one.
 *
 * basq_blocksum_sq_geo_f64: basq_blocksum_sq_f64 over the part of this rank's shard that geo_mode selects, exactly as
 *   basq_blocksum_geo_f64 does for the kernel itself (1: the regular region [0, reg_hi), the only mode with residue classes;
 *   2: the rest [reg_hi, R); 3: everything; 4: the ragged remainder as a block of its own -- SOBER/_rchq.py:127-135).  kobs
 *   holds this rank's live candidates from its FIRST one (column 0 = local position 0); the kernel advances it with `cand`.
 * basq_cov_diag_geo_f64: basq_cov_diag_f64 for the shard the descriptor names; the launch covers R_max candidates (an upper
 *   bound of the shard's length), out[p] is written for p below the actual length only.
 * basq_sq_noise_part_geo_f64: what those values contribute to the round's message (BASQ/_gp.py:275-276 inside
 *   _wsabi.py:240-242; BASQ/_rchq.py:88: U_svd @ ...), one work-group:
 *       dvec[s] = sum over this rank's candidates p of set s in FULL blocks of mu[p] val[p]      (position order)
 *       dt[k]   = mu val of remainder point k (Nystrom row k; k < m)
 *       part[0][.] = 0,   part[1 + r][s] = U[r][s] dvec[s]  (s < min(m, S))  +  [s == S - 1] sum_k U[r][k0 + k] dt[k]
 *                                          (+ sober != 0: U[r][s] dt[s - k0] for the remainder's first count, s in [k0, k1))
 *   U [q, ldu >= m] = the UNSCALED Nystrom basis (the squared term carries no warped means), part [rows >= q + 1, S] is written
 *   whole (rows beyond q zero), S <= 1024.  dvec is formed by several work-groups (fixed partition of the blocks, partial sums
 *   added in index order) in the caller's workspace ws [basq_sq_noise_part_ws_doubles(S)].
 */
int basq_blocksum_sq_geo_f64(const basq_kernel_spec* spec, const double* nys, int32_t m, const double* cand,
                             const double* mu, const int64_t* geo, int32_t geo_mode, int32_t S, int32_t n_chunks,
                             int32_t class_mod, int32_t class0, const double* bmatT, int64_t ldb, const double* kobs,
                             int64_t ldk, int32_t n_obs, double noise, double* Epart, void* stream);
int basq_cov_diag_geo_f64(const basq_kernel_spec* spec, const double* nys, int32_t m, const double* cand,
                          const int64_t* geo, int64_t R_max, int32_t S, const double* bmatT, int64_t ldb, const double* kobs,
                          int64_t ldk, int32_t n_obs, double noise, double* out, void* stream);
int64_t basq_sq_noise_part_ws_doubles(int32_t S);      /* size of basq_sq_noise_part_geo_f64's workspace `ws` (doubles) */
int basq_sq_noise_part_geo_f64(const double* mu, const double* val, const int64_t* geo, const double* U, int64_t ldu,
                               int32_t q, int32_t m, int32_t S, int32_t rows, int32_t sober, double* ws, double* part,
                               void* stream);

/*
 * Gaussian test matrix of torch.svd_lowrank (BASQ/_rchq.py:29 -> torch._lowrank: R = torch.randn(m, q)): the
 * Box-Muller half of torch's CPU normal_fill.  u [n] are the uniforms torch.rand(n, float64) draws from the CPU
 * generator (the same mt19937 consumption as torch.randn(n)); out[16b + j] = sqrt(-2 log(1 - u[16b+j])) *
 * cos(2 pi u[16b+j+8]), out[16b+j+8] = ... sin(...), j < 8.  If n % 16 != 0, u_tail [16] (16 further draws)
 * regenerates the last 16 outputs, as torch does; otherwise u_tail must be NULL.  n >= 16.
 */
int basq_box_muller_f64(const double* u, int64_t n, const double* u_tail, double* out, void* stream);

/*
 * CholeskyQR building block of the randomised range finder behind torch.svd_lowrank (BASQ/_rchq.py:29):
 * G [q,q] (symmetric positive definite, = X^T X) is overwritten by its Cholesky factor L (lower triangle;
 * the strict upper triangle is left as it was) and W [q,q] receives L^{-T} (upper triangular), so that
 * Q = X W has orthonormal columns.  info[0] = 0 on success, j+1 if pivot j <= rel_tol * max_i G[i][i]
 * (numerically rank-deficient: the caller falls back to Householder QR on the host).  q <= 1024.
 * W == NULL: factor only (G -> L), through a packed-triangle LDS kernel that also serves 142 < q <= 200, where the
 * square no longer fits in LDS; returns BASQ_EUNSUPPORTED for q > 200 (pass W then: global-memory kernel).
 */
int basq_chol_inv_f64(double* G, int32_t q, double* W, int32_t* info, double rel_tol, void* stream);

/*
 * CholeskyQR without an inverse (same place in the range finder as basq_chol_inv_f64, BASQ/_rchq.py:29):
 *   basq_chol_factor_f64: G [q,q] (SPD) -> its Cholesky factor L in the lower triangle (strict upper triangle left as it
 *       was), by a panel algorithm with the packed triangle in LDS: q/8 synchronised steps instead of q.  info[0] as for
 *       basq_chol_inv_f64.  q <= 200 (BASQ_EUNSUPPORTED beyond).
 *   basq_trsm_rows_f64:   Q [rows, q] = X [rows, q] L^-T  (row strides ldx / ldq; X == Q allowed), L as left by
 *       basq_chol_factor_f64: the orthonormal factor of CholeskyQR straight from X, 64 rows per work-group.  q <= 318.
 */
int basq_chol_factor_f64(double* G, int32_t q, int32_t* info, double rel_tol, void* stream);
/*   basq_cholqr_f64: the two above in ONE launch (q <= 200, rows <= 262144): G -> L in place and Q = X L^-T, the solve of
 *       column panel p starting as soon as the factor has produced it (work-group 0 factors and publishes, the others
 *       solve 64 rows each).  Same arithmetic and bits as basq_chol_factor_f64 + basq_trsm_rows_f64.  info: device
 *       int32[2] -- info[0] the pivot flag as above (or q + 1000: a solver gave up waiting for the factor), info[1] the
 *       progress word of the hand-over (zeroed by the entry; not for the caller). */
int basq_cholqr_f64(double* G, int32_t q, int32_t* info, double rel_tol, const double* X, int64_t ldx, int64_t rows,
                    double* Q, int64_t ldq, void* stream);
int basq_trsm_rows_f64(const double* X, int64_t ldx, int64_t rows, int32_t q, const double* L, double* Q, int64_t ldq,
                       void* stream);

/*
 * The big products of the randomised range finder (torch.svd_lowrank -> torch._lowrank.get_approximate_basis,
 * BASQ/_rchq.py:29: A @ R, A^H @ Q, A @ Q, Q^H @ A, and the Gram products X^T X of the orthonormalisations) as a
 * tall-skinny f64 GEMM on v_mfma_f64_4x4x4_4b:  C[M, N] = op(A) @ B[K, N],  op(A) = A[M, K] (trans = 0) or A^T with A
 * stored [K, M] (trans != 0), all row-major, N <= 208, ldb <= 2^24.  K is split into `ksplit` slices of whole 16-k trips;
 * their partial products go to work [>= ksplit * M * N] and are added in slice order (work may be NULL when one slice
 * results).  C is dense (ld = N).  B is only ever read inside its K x ldb extent (the kernel's wide row reads stop short of
 * the last rows, which are read masked).
 */
int basq_skinny_gemm_f64(const double* A, int64_t lda, int32_t trans, int32_t M, int32_t K, const double* B, int64_t ldb,
                         int32_t N, int32_t ksplit, double* work, double* C, void* stream);

/* Dense f64 GEMM on the matrix cores: C[M,N] = alpha * A[M,K] @ B[K,N] (row-major, lda/ldb/ldc). */
int basq_gemm_f64(const double* A, int64_t lda, const double* B, int64_t ldb, double* C, int64_t ldc, int32_t M,
                  int32_t N, int32_t K, double alpha, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BASQ_HIP_H */
