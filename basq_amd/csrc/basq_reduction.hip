// basq_reduction.hip -- the per-round reductions and the round bookkeeping: Caratheodory elimination, the null space from
// the bidiagonalisation's reflectors, re-weighting + compaction, round descriptors, class regrouping; and their C-ABI entries.
#include "basq_common.hpp"


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
#define BASQ_PAIRCOL(k) (2 * lane + ((k) & 1) + 128 * ((k) >> 1))
#define BASQ_WPG 8                        // waves per work-group of the cluster kernels: two per SIMD, 256 VGPRs each -- the
                                          // 100 x 200 matrix is 56 doubles per lane and stays in directly addressable VGPRs
                                          // (4 waves of 512 registers would park half of it in AGPRs: two moves per use)

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
                                                              double* __restrict__ PhiT, int xcd_stride) {
    // (round 6) only every xcd_stride-th work-group of the launch works: work-groups b, b + 8, ... are dealt to ONE XCD, the one
    // whose L2 holds the reflector rows the single-work-group bidiagonalisation has just written (and the null vectors for the
    // single-work-group elimination that follows) -- the launch was bound by fetching those rows across the fabric
    if (blockIdx.x % xcd_stride) return;
    // Reflector rows in flight.  V was written a moment ago by ONE work-group: for the other XCDs its lines come from
    // memory, not from their L2 (~1.2 us), i.e. four reductions' worth of latency is not enough -- 4 rows: 31.3 us,
    // 8: 28.1, 12: 25.7, 16: 25.6 (profiles/r02_m_nullspace_apply_rows_in_flight.txt)
    constexpr int PF = 12;
    const int lane = threadIdx.x & 63;
    const int c0 = (blockIdx.x / xcd_stride) * 4 + (threadIdx.x >> 6);
    if (c0 >= n - m) return;                              // wave-uniform
    double y[NV], v[PF][NV], tv[PF];
    // loads without branches (round 5: the conditional loads compiled into one exec-mask branch each): a row index below 0 is
    // clamped and gets tau = 0 (the identity, whatever its entries), a column beyond n is clamped and masked by a select
    int cc[NV];
    bool cok[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c = lane + 64 * k;
        cok[k] = c < n;
        cc[k] = cok[k] ? c : (n - 1);
        y[k] = (c == m + c0) ? 1.0 : 0.0;
    }
    auto load_row = [&](int p, int i) {
        const int ic = (i >= 0) ? i : 0;
        const double t = tau[ic];
        tv[p] = (i >= 0) ? t : 0.0;
        const double* row = V + (size_t)ic * n;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const double x = row[cc[k]];
            v[p][k] = cok[k] ? x : 0.0;
        }
    };
#pragma unroll
    for (int p = 0; p < PF; ++p) load_row(p, m - 1 - p);
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
#pragma unroll
            for (int k = 0; k < NV; ++k) y[k] -= t * v[p][k];
            load_row(p, i - PF);                          // refill the slot behind the reduction
        }
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c = lane + 64 * k;
        if (c < n) PhiT[(size_t)c0 * n + c] = y[k];
    }
}

// The same apply with 16-BYTE loads (round 6): lane l holds the column PAIRS 2l, 2l + 1 (+ 128 per further slot pair), so a reflector
// row is NV / 2 load instructions instead of NV.  Why it matters: the rows were written by the bidiagonalisation's launch a
// moment ago and come back from memory (the L2s are written back and invalidated at the kernel boundary; ~2 us), so the
// loop needs rows in flight -- and a wave can have 63 vector loads outstanding: at M > 256 (eight column slots) twelve rows of
// eight 8-byte loads exceed that, twelve rows of four 16-byte loads do not (752 -> 740 us at 200 x 400, profiles/r08_o_*).
// At 100 x 200 none of the three things tried moved the apply's 25 us: 24 rows in flight instead of 12 (this kernel), the
// work-groups on the bidiagonalisation's XCD (-3 us), four reflectors per reduction (compact-WY blocks: the four-fold wave
// sum costs what four single ones do; profiles/r08_m_*, r08_n_*).  Needs an even n (row starts 16-byte aligned).
template <int NV, int PF>
__global__ void __launch_bounds__(256) nullspace_apply_pairs_kernel(const double* __restrict__ V,
                                                                    const double* __restrict__ tau, int m, int n,
                                                                    double* __restrict__ PhiT, int xcd_stride) {
    static_assert(NV % 2 == 0, "column pairs");
    typedef double d2 __attribute__((ext_vector_type(2)));
    if (blockIdx.x % xcd_stride) return;
    const int lane = threadIdx.x & 63;
    const int c0 = (blockIdx.x / xcd_stride) * 4 + (threadIdx.x >> 6);
    if (c0 >= n - m) return;                              // wave-uniform
    constexpr int NP = NV / 2;
    double y[NV], v[PF][NV], tv[PF];
    int cc[NP];                                           // first column of pair h (clamped inside the row), n even: a pair is in or out
    bool cok[NP];
#pragma unroll
    for (int h = 0; h < NP; ++h) {
        const int c = 2 * lane + 128 * h;
        cok[h] = c < n;
        cc[h] = cok[h] ? c : (n - 2);
        y[2 * h] = (c == m + c0) ? 1.0 : 0.0;
        y[2 * h + 1] = (c + 1 == m + c0) ? 1.0 : 0.0;
    }
    auto load_row = [&](int p, int i) {
        const int ic = (i >= 0) ? i : 0;
        const double t = tau[ic];
        tv[p] = (i >= 0) ? t : 0.0;
        const double* row = V + (size_t)ic * n;
#pragma unroll
        for (int h = 0; h < NP; ++h) {
            const d2 x = *reinterpret_cast<const d2*>(row + cc[h]);
            v[p][2 * h] = cok[h] ? x.x : 0.0;
            v[p][2 * h + 1] = cok[h] ? x.y : 0.0;
        }
    };
#pragma unroll
    for (int p = 0; p < PF; ++p) load_row(p, m - 1 - p);
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
#pragma unroll
            for (int k = 0; k < NV; ++k) y[k] -= t * v[p][k];
            load_row(p, i - PF);                          // refill the slot behind the reduction
        }
    }
#pragma unroll
    for (int h = 0; h < NP; ++h) {
        const int c = 2 * lane + 128 * h;
        if (c < n) {
            d2 o;
            o.x = y[2 * h];
            o.y = y[2 * h + 1];
            *reinterpret_cast<d2*>(PhiT + (size_t)c0 * n + c) = o;
        }
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
// Epochs without a pairwise evaluation inside (round 6).  The candidates the residue classes do not cover -- the e < C full
// blocks behind the regular region and the ragged tail of t < S points -- obey the same law as the classes: a round sends the
// survivor of (block b, kept rank k) to position b * n_keep + k with its weight rescaled by w*_k / tot, and the tail, if set
// S - 1 is kept, follows behind.  Their contributions to a round's message are kept as MESSAGE COLUMNS, one per candidate:
// slot b < e holds block b (column j = the block's point in set j), one more slot holds the tail (column k = tail point k).
// The regular survivors fill exactly the next regular region (reg_blocks * n_keep = (reg_blocks / 2) * S), so the columns
// form a closed system: next round's columns are a gather + rescale of this round's -- no candidate is touched, nothing is
// evaluated or projected again until the next epoch starts.
//
// Buffer of a round ("parts"): [C class messages | 1 fold slot | E block slots | 1 tail slot], each [rows, S].  The fold slot
// is what the finalize kernel adds behind the classes: sum over the blocks, in index order, + the tail's columns, in index
// order, into set S - 1 (BASQ/_rchq.py:91-99).  In an epoch's FIRST round it is the message of the ordinary irregular chunk
// (basq_blocksum_geo_f64 mode 2, projected with the classes); the columns of that round are evaluated per block (mode 5) and
// per tail point (mode 4) beside the round's chain of single-work-group kernels, on another stream.
// ------------------------------------------------------------------------------------------------
struct IrrGeo {
    long long e, t;           // this round: full blocks behind the regular region, tail points
    long long e_n, t_n;       // next round (after the elimination's outcome)
    int n_keep, kr_last;      // kept sets; rank of set S - 1 among them (-1: the tail dies)
    bool bad;                 // violation / outcome the host did not size for: nothing may be derived from it
};

__device__ __forceinline__ IrrGeo irr_geometry(const long long* __restrict__ gp, const int* __restrict__ info,
                                               const int* __restrict__ keep_rank, int S, int E_in, int E_out) {
    IrrGeo g;
    const long long nb = gp[4], reg_blocks = gp[2] / S;
    g.e = nb - reg_blocks;
    g.t = gp[5];
    g.n_keep = info ? info[0] : 0;
    g.kr_last = keep_rank ? keep_rank[S - 1] : -1;
    g.bad = gp[3] != 0 || g.e < 0 || g.e > E_in || g.t < 0 || g.t >= S;
    if (info) g.bad = g.bad || info[1] != 0 || 2 * g.n_keep != S;
    const long long n_irr = g.bad ? 0 : g.e * g.n_keep + ((g.kr_last >= 0) ? g.t : 0);
    g.e_n = n_irr / S;
    g.t_n = n_irr - g.e_n * S;
    if (g.e_n > E_out) g.bad = true;
    return g;
}

// One launch behind every elimination inside an epoch: next round's class messages (regroup_classes_kernel), next round's
// irregular columns + their fold slot, next round's descriptor (round_next_body).  All read the elimination's outcome, none
// reads another's output.
__global__ void epoch_turn_kernel(const double* __restrict__ Pin, int C, int E_in, double* __restrict__ Pout, int E_out,
                                  int rows, int S, const int* __restrict__ kept, const int* __restrict__ keep_rank,
                                  const double* __restrict__ w_star, const double* __restrict__ tot,
                                  const int* __restrict__ info, const long long* __restrict__ gp, long long* __restrict__ gn,
                                  int nb_cls, int nb_irr, int nb_fold) {
#pragma clang fp contract(off)
    const long long stride = (long long)rows * S;
    int blk = (int)blockIdx.x;
    if (blk < nb_cls) {                                             // ---- classes: C -> C / 2
        const long long e = (long long)blk * blockDim.x + threadIdx.x;
        if (e >= (long long)(C / 2) * stride) return;
        const int sp = (int)(e % S);
        const long long rest = e / S;
        const int j = (int)(rest % rows), cp = (int)(rest / rows);
        const int H = S / 2, par = sp / H, k = sp - par * H;
        const int s = kept[k];
        if ((unsigned)s >= (unsigned)S) {
            Pout[e] = 0.0;
            return;
        }
        const double v = Pin[((long long)(2 * cp + par) * rows + j) * S + s];
        Pout[e] = (v * w_star[k]) / tot[s];                         // the order of BASQ/_rchq.py:113-114
        return;
    }
    blk -= nb_cls;
    if (blk >= nb_irr + nb_fold) {                                  // ---- the next round's descriptor
        if (threadIdx.x < 64) round_next_body(threadIdx.x, gp, info, keep_rank, S, -1, 1, gn);
        return;
    }
    const IrrGeo g = irr_geometry(gp, info, keep_rank, S, E_in, E_out);
    const double* Iin = Pin + (long long)(C + 1) * stride;         // block slots, then the tail slot at E_in
    double* fold_out = Pout + (long long)(C / 2) * stride;
    double* Iout = fold_out + stride;                               // block slots, then the tail slot at E_out
    // the column that lands at irregular index i (next round), row j: a kept column of a block, or a tail column
    auto column = [&](long long i, int j) -> double {
        const long long from_blocks = g.e * g.n_keep;
        if (i < from_blocks) {
            const long long b = i / g.n_keep;
            const int k = (int)(i - b * g.n_keep);
            const int s = kept[k];
            const double v = Iin[b * stride + (long long)j * S + s];
            return (v * w_star[k]) / tot[s];
        }
        const double v = Iin[(long long)E_in * stride + (long long)j * S + (i - from_blocks)];
        return (v * w_star[g.kr_last]) / tot[S - 1];
    };
    if (blk < nb_irr) {                                             // ---- next round's columns
        const long long e = (long long)blk * blockDim.x + threadIdx.x;
        if (e >= (long long)(E_out + 1) * stride) return;
        const int sp = (int)(e % S);
        const long long rest = e / S;
        const int j = (int)(rest % rows);
        const long long slot = rest / rows;
        double v = 0.0;
        if (!g.bad) {
            if (slot < E_out) {
                if (slot < g.e_n) v = column(slot * S + sp, j);
            } else if (sp < g.t_n) {
                v = column(g.e_n * S + sp, j);
            }
        }
        Iout[e] = v;
        return;
    }
    blk -= nb_irr;                                                  // ---- their fold slot
    const int n_plain = (rows * S + (int)blockDim.x - 1) / (int)blockDim.x;
    if (blk < n_plain) {                                            // every entry but column S - 1: the blocks, in index order
        const int idx = blk * blockDim.x + threadIdx.x;
        if (idx >= rows * S) return;
        const int sp = idx % S, j = idx / S;
        if (sp == S - 1) return;
        double v = 0.0;
        if (!g.bad)
            for (long long b = 0; b < g.e_n; ++b) v += column(b * S + sp, j);
        fold_out[idx] = v;
        return;
    }
    // column S - 1 (BASQ/_rchq.py:91-99: the ragged tail belongs to the last set): one WAVE per row -- up to S - 1 tail columns
    // go into this one entry, and a single lane walking them was the launch's duration (33 us for 100 columns).  Fixed order:
    // lane l adds its columns k = l, l + 64, ... in that order, the 64 partial sums meet in an xor butterfly (offsets 32 .. 1;
    // the same bits in every lane), and the total is added behind the blocks' sum.
    const int lane = threadIdx.x & 63;
    const int j = (blk - n_plain) * ((int)blockDim.x >> 6) + ((int)threadIdx.x >> 6);
    if (j >= rows) return;
    double part = 0.0;
    if (!g.bad)
        for (long long k = lane; k < g.t_n; k += 64) part += column(g.e_n * S + k, j);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) part += __shfl_xor(part, o, 64);
    if (lane == 0) {
        double v = 0.0;
        if (!g.bad)
            for (long long b = 0; b < g.e_n; ++b) v += column(b * S + (S - 1), j);
        fold_out[(long long)j * S + (S - 1)] = v + part;
    }
}

// Survivor re-weighting + compaction of SEVERAL rounds at once (the rounds of an epoch touch no candidate: their compactions
// are applied together when the next epoch -- or the host's round-by-round loop -- needs the candidates again).  Per round the
// arithmetic of reweight_compact_geo_kernel, in round order: mu <- (mu * w*) / tot.
#define BASQ_MAX_EPOCH_ROUNDS 8
struct CompactRounds {
    const int* keep_rank[BASQ_MAX_EPOCH_ROUNDS];
    const double* w_star[BASQ_MAX_EPOCH_ROUNDS];
    const double* tot[BASQ_MAX_EPOCH_ROUNDS];
    const int* info[BASQ_MAX_EPOCH_ROUNDS];
    int n;
};
__global__ void __launch_bounds__(256) reweight_compact_rounds_kernel(
    const double* __restrict__ cand, const double* __restrict__ mu, const long long* __restrict__ gid,
    const double* __restrict__ wx, const long long* __restrict__ geo, const CompactRounds Rr, int S, int kp, long long out_rows,
    int expect_keep, double* __restrict__ cand_out, double* __restrict__ mu_out, long long* __restrict__ gid_out,
    double* __restrict__ wx_out) {
#pragma clang fp contract(off)
    // ONE lane per candidate walks the rounds (a few table look-ups each; all but 2^-rounds of the candidates drop out on the
    // way); the survivors of a wave -- two of 64 after five rounds -- then have their packed rows copied by the whole wave.
    // this rank's shard of the first round: local candidate p is global position off0 + p (one rank: [0, R))
    const long long off0 = geo[6], Rl0 = geo[7];
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    long long pg = (p < Rl0) ? (off0 + p) : -1;
    double w = (pg >= 0) ? mu[p] : 0.0;
    for (int r = 0; r < Rr.n; ++r) {
        const long long* g = geo + 8 * r;
        const long long n_full = g[1];
        const int n_keep = Rr.info[r][0];
        if (g[3] != 0 || Rr.info[r][1] != 0 || (expect_keep >= 0 && n_keep != expect_keep)) return;   // uniform: the host repeats the rounds
        if (pg < 0) continue;
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
        const int kr = Rr.keep_rank[r][set];
        if (kr < 0) {
            pg = -1;
            continue;
        }
        if (pg < n_full) dst += kr;
        const double scaled = w * Rr.w_star[r][kr];                            // :113-114 / :121-122
        w = scaled / Rr.tot[r][set];
        pg = dst;
    }
    if (pg >= 0) pg -= geo[8 * Rr.n + 6];                                       // local index in the shard of the round that follows
    if (pg >= out_rows) pg = -1;                                               // never outside the caller's buffers
    if (pg >= 0) {
        mu_out[pg] = w;
        gid_out[pg] = gid[p];
        if (wx) wx_out[pg] = wx[p];
    }
    unsigned long long alive = __ballot(pg >= 0);
    while (alive) {
        const int src_lane = (int)__builtin_ctzll(alive);
        alive &= alive - 1;
        const long long dst = __shfl(pg, src_lane, 64);
        const long long src = p - lane + src_lane;
        for (int k = lane; k < kp; k += 64) cand_out[dst * kp + k] = cand[src * kp + k];
    }
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

int basq_epoch_turn_f64(const double* Pin, int32_t C, int32_t E_in, double* Pout, int32_t E_out, int32_t rows, int32_t S,
                        const int32_t* kept, const int32_t* keep_rank, const double* w_star, const double* tot,
                        const int32_t* info, const int64_t* geo, int64_t* geo_next, void* stream) {
    if (!Pin || !Pout || !kept || !keep_rank || !w_star || !tot || !info || !geo || !geo_next) return BASQ_EINVAL;
    if (C < 2 || (C & 1) || E_in < 0 || E_out < 0 || rows < 1 || S < 2 || (S & 1)) return BASQ_EINVAL;
    const long long stride = (long long)rows * S;
    const long long nb_cls = ((long long)(C / 2) * stride + 255) / 256, nb_irr = ((long long)(E_out + 1) * stride + 255) / 256,
                    nb_fold = (stride + 255) / 256 + (rows + 3) / 4;       // + one wave per row for column S - 1
    if (nb_cls + nb_irr + nb_fold + 1 > 0x7fffffffLL) return BASQ_EINVAL;
    hipLaunchKernelGGL(epoch_turn_kernel, dim3((unsigned)(nb_cls + nb_irr + nb_fold + 1)), dim3(256), 0, (hipStream_t)stream, Pin, C,
                       E_in, Pout, E_out, rows, S, kept, keep_rank, w_star, tot, info, (const long long*)geo, (long long*)geo_next,
                       (int)nb_cls, (int)nb_irr, (int)nb_fold);
    BASQ_CHECK_LAUNCH();
    return BASQ_OK;
}

int basq_reweight_compact_rounds_f64(const double* cand, const double* mu, const int64_t* gid, const double* wx,
                                     const int64_t* geo, int32_t n_rounds, const int32_t* const* keep_rank,
                                     const double* const* w_star, const double* const* tot, const int32_t* const* info,
                                     int64_t R_max, int32_t S, int32_t kp, int64_t out_rows, int32_t expect_keep,
                                     double* cand_out, double* mu_out, int64_t* gid_out, double* wx_out, void* stream) {
    if (!cand || !mu || !gid || !geo || !keep_rank || !w_star || !tot || !info || !cand_out || !mu_out || !gid_out)
        return BASQ_EINVAL;
    if (n_rounds < 1 || n_rounds > BASQ_MAX_EPOCH_ROUNDS || R_max < 1 || S < 1 || kp < 1 || out_rows < 1 || (wx && !wx_out))
        return BASQ_EINVAL;
    CompactRounds Rr;
    Rr.n = n_rounds;
    for (int r = 0; r < BASQ_MAX_EPOCH_ROUNDS; ++r) {
        const int q = r < n_rounds ? r : 0;
        if (!keep_rank[q] || !w_star[q] || !tot[q] || !info[q]) return BASQ_EINVAL;
        Rr.keep_rank[r] = keep_rank[q]; Rr.w_star[r] = w_star[q]; Rr.tot[r] = tot[q]; Rr.info[r] = info[q];
    }
    const long long nt = (long long)R_max;                      // one lane per candidate
    hipLaunchKernelGGL(reweight_compact_rounds_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, (hipStream_t)stream, cand,
                       mu, (const long long*)gid, wx, (const long long*)geo, Rr, S, kp, (long long)out_rows, expect_keep, cand_out,
                       mu_out, (long long*)gid_out, wx_out);
    BASQ_CHECK_LAUNCH();
    return BASQ_OK;
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
    // BASQ_NS_APPLY_XCD=1: the launch's work-groups on ONE XCD (<= 32 of them) -- measured: 3 us of 259 at 100 x 200
    // (profiles/r08_n_*), not worth tying 25 work-groups to one XCD while other batches are in flight; default: all XCDs
    static const int apply_xcd = [] { const char* e = getenv("BASQ_NS_APPLY_XCD"); return e ? atoi(e) : 0; }();
    const int n_wg = (nvec + 3) / 4;
    const int xs = (apply_xcd && n_wg <= 32) ? cluster_stride() : 1;
    const dim3 grid((unsigned)(n_wg * xs)), block(256);
    // 16-byte loads (BASQ_NS_APPLY_PAIRS=0: the 8-byte form): even M, buffers 16-byte aligned.  Measured (profiles/r08_o_*):
    // 752 -> 740 us at 200 x 400, where twelve rows of eight 8-byte loads exceed a wave's 63 outstanding loads; nothing at
    // 100 x 200 (24 rows in flight instead of 12: 257 vs 256 us), which keeps the 8-byte kernel
    static const int apply_pairs = [] { const char* e = getenv("BASQ_NS_APPLY_PAIRS"); return e ? atoi(e) : 1; }();
    const bool pairs = apply_pairs && (M % 2 == 0) && (((uintptr_t)V | (uintptr_t)PhiT) % 16 == 0);
    if (pairs && M > 256 && M <= 512) hipLaunchKernelGGL((nullspace_apply_pairs_kernel<8, 12>), grid, block, 0, st, V, tau, s, M, PhiT, xs);
    else if (M <= 256) hipLaunchKernelGGL(nullspace_apply_kernel<4>, grid, block, 0, st, V, tau, s, M, PhiT, xs);
    else if (M <= 512) hipLaunchKernelGGL(nullspace_apply_kernel<8>, grid, block, 0, st, V, tau, s, M, PhiT, xs);
    else hipLaunchKernelGGL(nullspace_apply_kernel<16>, grid, block, 0, st, V, tau, s, M, PhiT, xs);
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

}  // extern "C"
