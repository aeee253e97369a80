"""Recombination engine: the host orchestration of the HIP kernels (one process per GPU).

Mirrors ``BASQ/_rchq.py``: ``rc_kernel_svd`` (:34-40) = Nystrom basis + ``Mod_Tchernychova_Lyons``
(:43-130) whose per-round reduction is ``Tchernychova_Lyons_CAR`` (:133-175).  What runs where:

=======================================  ==========================================================
step (reference lines)                   here
=======================================  ==========================================================
Gram ``kernel(pt, pt)`` (:29)            ``basq_gram_f64`` (+ two small library GEMMs per batch for GP corrections)
``torch.svd_lowrank`` (:29)              same algorithm (Halko 4.4/5.1, niter=2): uniforms of the Gaussian test
                                         matrix from the CPU global generator (parity) + Box-Muller on the GPU,
                                         the products on ``basq_skinny_gemm_f64``, CholeskyQR
                                         (``basq_chol_factor_f64`` + ``basq_trsm_rows_f64``) instead of the
                                         Householder QRs, one q x q SVD on host LAPACK
hot loop + tail + tot (:79-99)           ``basq_blocksum_f64`` (fused, nothing materialised; per residue class of
                                         the block index, so that the next rounds regroup instead of re-evaluating:
                                         ``basq_regroup_classes_f64``)
round geometry (:76-78, :107-130)        closed form, on the device: ``basq_round_next_i64`` + the ``*_geo`` entries
                                         (one rank: no host wait per round)
``U_svd @ X_for_nys`` (:88)              ``basq_project_f64`` (f64 MFMA)
divide, ones column (:101, :138)         ``basq_finalize_f64``
full SVD -> null space (:140-143)        ``basq_nullspace_f64``: the right Householder reflectors of gesdd's
                                         bidiagonal reduction, i.e. LAPACK's own null-space rows (host
                                         ``torch.linalg.svd`` kept behind ``GPU_NULLSPACE = False``): the basis
                                         is algorithm-specific, any other orthonormal null-space basis
                                         changes the selection (SURVEY finding 3)
elimination loop (:146-175)              ``basq_car_eliminate_f64``
re-weight + compaction (:107-130)        ``basq_reweight_compact_f64`` (closed-form destinations)
=======================================  ==========================================================

Multi-GPU: the candidate pool is sharded in contiguous id ranges; per round every rank
block-sums and projects its shard, the ``(q+1) x S`` messages are all-gathered and added in rank order, and
every rank runs the (deterministic) reduction on the same message; re-weighting/compaction are local.  The range
finder's Gram products are row-sharded as well (``_ShardedProducts``).
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field

import torch

from ._lib import ROLE_A, ROLE_B
from ._partition import (RoundGeometry, choose_chunks, initial_shards, local_blocks, next_shard,
                         survivors_before)


# ----------------------------------------------------------------------------------------------------
# communicators
# ----------------------------------------------------------------------------------------------------
class LocalComm:
    rank, world = 0, 1

    def all_gather(self, t):
        return t.unsqueeze(0)

    def broadcast(self, t, src=0):
        return t


class TorchDistComm:
    """``torch.distributed`` (backend ``nccl`` = RCCL over xGMI on ROCm; ``gloo`` in the CPU tests)."""

    def __init__(self, group=None):
        import torch.distributed as dist

        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    def all_gather(self, t):
        """-> ``[world, *t.shape]`` (rank order).  Messages are tiny ((q+1) x S doubles): latency-bound."""
        out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        # ONE flat output buffer (RCCL: a single ncclAllGather, no per-rank output list / copy-out kernels)
        self.dist.all_gather_into_tensor(out.view(-1), t.contiguous().view(-1), group=self.group)
        return out

    def broadcast(self, t, src=0):
        self.dist.broadcast(t, src=src, group=self.group)
        return t


# ----------------------------------------------------------------------------------------------------
# trace (tests / profiling)
# ----------------------------------------------------------------------------------------------------
@dataclass
class EngineTrace:
    rounds: list = field(default_factory=list)      # dicts: R, S, nb, n_tail, kept, tot, XcarT (optional)
    U: torch.Tensor | None = None
    timers: dict = field(default_factory=dict)
    keep_tensors: bool = False
    time_kernels: bool = False                      # record HIP events around every block-sum launch
    kernel_events: list = field(default_factory=list)   # (start_evt, end_evt, dict(pairs=, R=, m=, S=))
    host_sync: bool = True                          # synchronise around phases to attribute host timers

    def add_time(self, key, dt):
        self.timers[key] = self.timers.get(key, 0.0) + dt


class _Timer:
    """Host timer feeding ``EngineTrace.timers`` (synchronising only when the trace asks for it)."""

    def __init__(self, ops, trace, key, sync=True):
        self.ops, self.trace, self.key, self.sync = ops, trace, key, sync

    def __enter__(self):
        if self.trace is not None:
            if self.sync and self.trace.host_sync:
                self.ops.synchronize()
            self.t0 = time.perf_counter()
        return self

    def __exit__(self, *exc):
        if self.trace is not None:
            if self.sync and self.trace.host_sync:
                self.ops.synchronize()
            self.trace.add_time(self.key, time.perf_counter() - self.t0)
        return False


# Host LAPACK calls on small matrices (100 x 200 SVD, 1e4 x 99 QR) are slower, not faster, on a many-core
# host with every core in the team (measured: 30 ms per 100x200 gesdd with 128 threads vs ~2 ms with 8).
HOST_LAPACK_THREADS = 8          # tall QR of the fallback path
HOST_SVD_THREADS = 1             # 100x200 / 99x99 SVDs: fastest single-threaded (profiles/r01_host_lapack_threads.txt)


class _lapack_threads:
    def __init__(self, n=None):
        self.n = n or HOST_LAPACK_THREADS

    def __enter__(self):
        self.prev = torch.get_num_threads()
        if self.prev > self.n:
            torch.set_num_threads(self.n)
        return self

    def __exit__(self, *exc):
        if torch.get_num_threads() != self.prev:
            torch.set_num_threads(self.prev)
        return False


def _host_qr_q(ops, X, trace=None):
    """Q factor by host LAPACK (geqrf/orgqr), as ``torch.linalg.qr(X).Q`` on a CPU tensor."""
    with _Timer(ops, trace, "basis.d2h"):
        Xh = X.cpu()
    with _Timer(ops, trace, "basis.host_qr"), _lapack_threads():
        Qh = torch.linalg.qr(Xh).Q
    with _Timer(ops, trace, "basis.h2d"):
        return ops.to_device(Qh)


def _splits_for(K: int, want: int) -> int:
    for c in range(min(want, K), 0, -1):
        if K % c == 0:
            return c
    return 1


# The range finder's products on basq_skinny_gemm_f64 (False: library GEMMs through torch, for A/B comparisons).
OWN_RANGE_GEMM = True


def _mm_splitk(ops, A, B, want: int = 8):
    """``A @ B`` for a long contraction with a skinny output (<= 208 columns): the hand-written tall-skinny MFMA kernel
    (``basq_skinny_gemm_f64``).  Wider outputs, other dtypes and the CPU stand-in take the library path below: ONE
    batched library GEMM over K slices.

    rocBLAS tiles the output only (no split-K): ``[1e4,1e4] @ [1e4,99]`` is 79 work-groups on 256 CUs and
    ``[99,1e4] @ [1e4,99]`` a single one.  Viewing the K dimension as (splits, K/splits) -- strided views, no
    copies -- runs ``splits`` times more work-groups concurrently; the partial products are added in slice order.
    """
    sk = getattr(ops, "skinny_gemm", None) if OWN_RANGE_GEMM else None
    if sk is not None and A.dim() == 2 and B.dim() == 2 and B.shape[1] <= ops.SKINNY_MAX_N and B.stride(1) == 1 \
            and A.dtype == torch.float64 and B.dtype == torch.float64:
        # the hand-written tall-skinny MFMA kernel (basq_skinny_gemm_f64): A read once, split-K inside
        if A.stride(1) == 1 and A.stride(0) >= A.shape[1]:
            return sk(A, B, False)
        if A.stride(0) == 1 and A.stride(1) >= A.shape[0]:
            return sk(A.t(), B, True)                          # A is a transposed view: read the stored matrix
    M, K = A.shape
    N = B.shape[1]
    c = _splits_for(K, want)
    if c == 1 or not hasattr(ops, "device") or getattr(ops, "name", "") != "hip":
        return ops.matmul(A, B)
    Ab = A.unflatten(1, (c, K // c)).permute(1, 0, 2)          # [c, M, K/c]  (view)
    Bb = B.unflatten(0, (c, K // c))                            # [c, K/c, N]  (view)
    return torch.bmm(Ab, Bb).sum(0)


def _cholqr(ops, X, flags, passes=2):
    """Basis of range(X) ([m,k], k <= m) by CholeskyQR passes, entirely on the GPU.

    ``X^T X = L L^T``, ``Q = X L^{-T}``; with two passes Q is orthonormal to round-off whenever
    cond(X) < ~1e7 (needed for the final basis); one pass (orthonormal to ~cond^2 eps, i.e. perfectly
    conditioned for the next multiplication by A) is enough for the intermediate subspace iterates, whose
    only role is their range.  The pivot flags (device int32) are appended to ``flags`` and checked once, later.

    ``Q = X L^{-T}`` is a row-parallel triangular solve (``basq_trsm_rows_f64``) against the factor of the panel Cholesky
    (``basq_chol_factor_f64``): no inverse is formed.  Ranks beyond the panel kernel's LDS capacity (k > 200) use the
    inverse-based kernels of round 1.
    """
    k = X.shape[1]
    for _ in range(passes):
        G = _mm_splitk(ops, X.t(), X, 32)
        if k <= getattr(ops, "CHOL_FACTOR_MAX_Q", 0):
            flags.append(ops.chol_factor(G))
            X = ops.trsm_rows(X, G)
        else:
            W, info = ops.chol_inv(G)
            flags.append(info)
            X = _mm_splitk(ops, X, W, 1)
    return X


def _gaussian_test_matrix(ops, m, q, trace=None):
    """``R = torch.randn(m, q)`` of ``torch._lowrank.get_approximate_basis`` with the reference's RNG consumption.

    torch's CPU ``normal_`` first fills the tensor with mt19937 uniforms and then applies Box-Muller in blocks of
    16 (scalar libm, ~12 ms for 1e4 x 99).  ``torch.rand`` makes exactly the same draws (the generator ends in
    the same state: ``test_rand_consumes_like_randn``), so only the uniforms are produced on the host -- straight
    into the pinned staging buffer, no intermediate allocation -- and the transform runs on the GPU; values agree
    with ``torch.randn`` to 1 ulp (device vs host libm), far inside the stability margin of the selection.
    """
    n = m * q
    if n < 16:
        return ops.to_device(torch.randn(m, q, dtype=torch.float64))
    with _Timer(ops, trace, "basis.rand_host", sync=False):
        u = ops.host_uniform(n, "rand_u")                        # CPU global generator
        ut = ops.host_uniform(16, "rand_ut") if n % 16 else None
    with _Timer(ops, trace, "basis.rand_h2d", sync=False):
        # (the 8-MB copy on a stream of its own, so that it does not queue behind the round-1 block sums, was tried: no gain
        # at the headline size, +2 ms at N = 1e5, where the host is not ahead of the GPU -- profiles/r02_l README entry)
        R = ops.box_muller(ops.from_pinned(u), None if ut is None else ops.from_pinned(ut))
    return R.view(m, q)


def _skip_test_matrix_draw(ops, m, q):
    """Advance the CPU global generator exactly as ``_gaussian_test_matrix(ops, m, q)`` does, without the device work.

    Multi-rank runs draw the Gaussian test matrix on rank 0 and broadcast it; the other ranks call this, so that every
    rank's global generator stays in lock-step with rank 0's -- a BASQ loop that samples its next pool from the global
    RNG on every rank (as ``bench.py`` and ``PriorSampler`` do) keeps seeing ONE pool, not one per rank."""
    n = m * q
    if n < 16:
        torch.randn(m, q, dtype=torch.float64)
        return
    ops.host_uniform(n, "rand_u")
    if n % 16:
        ops.host_uniform(16, "rand_ut")


# Multi-rank: every rank reduces the gathered message itself (the kernels sum in a fixed order, so all ranks obtain
# the same survivors bit for bit) instead of rank 0 reducing and broadcasting the result: one collective less per round.
REPLICATED_REDUCTION = True
SHARDED_BASIS = True             # multi-rank: split the range finder's Gram products over the ranks (False: rank 0 only)
LATE_CHUNKS = 1                  # chunks of the round-1 block sums deferred behind the range finder (0 = none)
LATE_CLASSES = 2                 # the same in class mode: the deferred classes (2 of 16 = 1.1 ms of GPU work) run while the
                                 # host does the range finder's q x q SVD; A/B on MI355X: 42.9 / 43.35 / 43.15 batches/s for 1 / 2 / 3
# Residue-class block sums: evaluate the pairwise kernel once per EPOCH of log2(C) + 1 rounds.  The chunks of the block
# sums are the residue classes of the block index modulo C; a round that keeps exactly half of the sets sends the
# survivor of (block b, kept rank k) to (block b // 2, set (b % 2) * n + k), so the next round's sums -- again per class,
# modulo C / 2 -- are a gather + rescale of this round's (``basq_regroup_classes_f64``): no candidate is touched.  Only
# the blocks beyond a multiple of C and the ragged tail (< (C + 1) * S points, halving every round) are evaluated directly.
BASIS_SIDE_STREAM = False        # one rank: range finder on a second stream beside the round-1 block sums (A/B on MI355X: 25.6 vs 24.6 ms -- off)
CLASS_SUMS = True
ASYNC_ROUNDS = True              # one rank, plain block sums: rounds driven by a device-resident descriptor, no host wait per round
MAX_CLASSES = 16                 # classes at the start of an epoch (power of two): 16 -> the kernel runs in rounds 1, 6, 11


def _classes_for(nb_global: int, m: int, S: int, kk: int) -> int:
    """Number of residue classes (a power of two, 1 = none) for an evaluation over ``nb_global`` full blocks."""
    if not CLASS_SUMS:
        return 1
    c = MAX_CLASSES
    while c > 1 and nb_global < 4 * c:           # at least four blocks per class (the classes are also the chunks)
        c //= 2
    return c


def _late_split(off: int, Rl: int, n_full: int, S: int, n_chunks: int, n_late: int):
    """Local position at which the round-1 block sums can be cut into two launches with UNCHANGED chunk boundaries
    (``basq_blocksum_f64`` splits the block range evenly: chunk c = blocks ``[lo + c*per, lo + (c+1)*per)``), or None.

    The first launch takes chunks ``0 .. n_chunks-n_late-1``, the second the rest (incl. the ragged tail, which
    belongs to the last chunk).  Both launches recompute ``per`` from their own ranges: the cut is only taken when
    they arrive at the same value, so that every partial sum is bit-identical to the single-launch result.
    """
    if n_late < 1 or n_chunks < 4 or n_late >= n_chunks:
        return None
    lim = min(off + Rl, n_full)
    if lim <= off:
        return None
    lo, hi = off // S, -(-lim // S)
    per = max(1, -(-(hi - lo) // n_chunks))
    c_a = n_chunks - n_late
    rest = (hi - lo) - c_a * per
    if rest < 1 or max(1, -(-rest // n_late)) != per:
        return None
    p = (lo + c_a * per) * S - off
    return p if 0 < p < Rl else None


# The GPU range finder may be switched off (tests compare both paths).
GPU_RANGE_FINDER = True
# Per-round null space (:140-143) from the bidiagonalisation's right reflectors on the GPU (basq_nullspace_f64)
# instead of a host LAPACK SVD; False restores the host path (same rows to ~1e-13, see tests).
GPU_NULLSPACE = True


class _DenseProducts:
    """The Nystrom Gram matrix ``A`` resident on this GPU: the three products of the range finder (``_mm_splitk``)."""

    def __init__(self, ops, A):
        self.ops, self.A, self.At, self.m = ops, A, A.t(), A.shape[0]

    def draw(self, q, trace):
        return _gaussian_test_matrix(self.ops, self.m, q, trace)

    def a(self, Q):
        return _mm_splitk(self.ops, self.A, Q)

    def at(self, Q):
        return _mm_splitk(self.ops, self.At, Q)

    def qta(self, Q):
        return _mm_splitk(self.ops, Q.t(), self.A)

    def full(self):
        return self.A


class _ShardedProducts:
    """Rows ``[r0, r0 + mr)`` of ``A`` on this rank (multi-GPU, SURVEY 8e: the range finder no longer idles W-1 GPUs).

    ``A`` is a kernel Gram matrix -- symmetric up to the rounding of its entries -- so ``A^T Q`` and ``(Q^T A)^T`` are
    computed as ``A Q`` as well: every product is ``A_rows @ Q`` on each rank followed by ONE all-gather of the
    ``[mr, q]`` blocks (1 MB per rank at the headline size), after which all ranks hold the same ``[m, q]`` matrix and
    run the small replicated steps (CholeskyQR, LQ, the q x q host SVD) identically.  Only rank 0 consumes the RNG: the
    Gaussian test matrix is broadcast.  The deviation from the single-GPU arithmetic (A for A^T) is at rounding
    level, far inside the stability margin of the selection (SURVEY finding 3); the gloo tests pin the indices.
    """

    def __init__(self, ops, comm, A_rows, shards, m):
        self.ops, self.comm, self.rows, self.shards, self.m = ops, comm, A_rows, shards, m
        self.mb = max(n for _, n in shards)

    def draw(self, q, trace):
        if self.comm.rank == 0:
            R = _gaussian_test_matrix(self.ops, self.m, q, trace).contiguous()
        else:
            _skip_test_matrix_draw(self.ops, self.m, q)         # same generator consumption on every rank
            R = self.ops.empty(self.m, q)
        return self.comm.broadcast(R)

    def a(self, Q):
        mr = self.rows.shape[0]
        blk = self.ops.zeros(self.mb, Q.shape[1])
        if mr:
            blk[:mr] = _mm_splitk(self.ops, self.rows, Q, 64)
        g = self.comm.all_gather(blk)                            # [W, mb, q]
        return torch.cat([g[r, :n] for r, (_, n) in enumerate(self.shards)], 0)

    at = a

    def qta(self, Q):
        return self.a(Q).t().contiguous()

    def full(self):
        blk = self.ops.zeros(self.mb, self.m)
        blk[:self.rows.shape[0]] = self.rows
        g = self.comm.all_gather(blk)
        return torch.cat([g[r, :n] for r, (_, n) in enumerate(self.shards)], 0)


def nystrom_basis(ops, A, q_req: int, trace: EngineTrace | None = None, overlap=None):
    """``ker_svd_sparsify`` (``BASQ/_rchq.py:28-31``): ``-svd_lowrank(A, q)[0].T`` -> ``[min(q,m), m]``.

    Restates ``torch._lowrank.get_approximate_basis`` / ``_svd_lowrank`` (torch 2.10, niter=2, square A so
    no transposition): the Gaussian test matrix is drawn exactly where the reference draws it -- one
    ``torch.randn(m, q)`` from the CPU global generator.

    What the reference's result depends on is only (i) that draw and (ii) the *range* of each intermediate
    ``Q``: the rows of the returned ``U`` are the left singular vectors of ``Q Q^T A``, unique up to sign,
    and the recombination is bit-for-bit invariant under row sign flips of ``U`` (tests/test_oracle.py).
    So the five Householder QRs (host LAPACK in the reference) are replaced by CholeskyQR2 on the GPU and
    the ``[k, m]`` SVD by an LQ reduction on the GPU + a ``k x k`` SVD on the host.  If a Cholesky pivot
    signals a numerically rank-deficient panel (cond > ~1e6) the whole basis is recomputed with host
    Householder QR, from the same Gaussian draw.

    ``A``: the Gram matrix (a tensor) or a products object (``_DenseProducts`` / ``_ShardedProducts``).
    ``overlap``: optional callable that enqueues independent GPU work; it is called once, right after the copy of
    the small ``L`` factor to the host has been enqueued, so that work runs while the host does the ``k x k`` SVD
    (otherwise ~1 ms of GPU idle time per batch).
    """
    prod = _DenseProducts(ops, A) if torch.is_tensor(A) else A
    m = prod.m
    with _Timer(ops, trace, "basis.randn"):
        R = prod.draw(q_req, trace)
    if GPU_RANGE_FINDER and q_req <= m:
        with _Timer(ops, trace, "basis.gpu_range"):
            flags = []
            Q = _cholqr(ops, prod.a(R), flags, passes=1)
            Q = _cholqr(ops, prod.at(Q), flags, passes=1)
            Q = _cholqr(ops, prod.a(Q), flags, passes=1)
            Q = _cholqr(ops, prod.at(Q), flags, passes=1)
            Q = _cholqr(ops, prod.a(Q), flags, passes=2)         # the basis that is actually used
            # LQ of B = Q^T A ([k, m]) by CholeskyQR2 on its rows, formed on Y = B^T = A^T Q ([m, k]: tall, row-parallel):
            #   B = L1 L2 Qb^T  ->  the left singular vectors of B are those of L = L1 L2
            Y = prod.at(Q)
            k = Y.shape[1]
            G1 = _mm_splitk(ops, Y.t(), Y, 32)                 # = B B^T
            if k <= getattr(ops, "CHOL_FACTOR_MAX_Q", 0):
                i1 = ops.chol_factor(G1)
                Yq = ops.trsm_rows(Y, G1)                       # = (L1^-1 B)^T
                G2 = _mm_splitk(ops, Yq.t(), Yq, 32)
                i2 = ops.chol_factor(G2)
            else:
                W1, i1 = ops.chol_inv(G1)
                Yq = _mm_splitk(ops, Y, W1, 1)
                G2 = _mm_splitk(ops, Yq.t(), Yq, 32)
                _, i2 = ops.chol_inv(G2)
            L = _mm_splitk(ops, torch.tril(G1), torch.tril(G2), 1)
            bad = torch.stack([f.reshape(()) for f in flags + [i1, i2]]).max()
            both, ready = ops.to_host_async(torch.cat([L.reshape(-1), bad.to(torch.float64).reshape(1)]), "basisL")
        if overlap is not None:
            overlap()
            overlap = None
        with _Timer(ops, trace, "basis.host_svd"):
            ready.synchronize()                                # one wait for the whole range finder
            Lh = both[:k * k].reshape(k, k)
            ok = int(both[k * k].item()) == 0
            if ok:
                with _lapack_threads(HOST_SVD_THREADS):
                    Ub = torch.linalg.svd(Lh)[0]
        if ok:
            with _Timer(ops, trace, "basis.gemm"):
                U = _mm_splitk(ops, Q, ops.to_device(Ub), 1)   # [m, k]
                return (-1 * U.t()).contiguous()               # :30
        if trace is not None:
            trace.timers["basis.fallback"] = trace.timers.get("basis.fallback", 0) + 1
    if overlap is not None:
        overlap()
    A = prod.full()                                            # (sharded: gathered -- the rare path)
    At = A.t()
    with _Timer(ops, trace, "basis.gemm"):
        X = ops.matmul(A, R)
    Q = _host_qr_q(ops, X, trace)
    for _ in range(2):
        with _Timer(ops, trace, "basis.gemm"):
            X = ops.matmul(At, Q)
        Q = _host_qr_q(ops, X, trace)
        with _Timer(ops, trace, "basis.gemm"):
            X = ops.matmul(A, Q)
        Q = _host_qr_q(ops, X, trace)
    with _Timer(ops, trace, "basis.gemm"):
        B = ops.matmul(Q.t(), A)                               # [k, m]
    with _Timer(ops, trace, "basis.d2h"):
        Bh = B.cpu()
    with _Timer(ops, trace, "basis.host_svd"), _lapack_threads():
        Ub, _, _ = torch.linalg.svd(Bh, full_matrices=False)
    with _Timer(ops, trace, "basis.gemm"):
        U = ops.matmul(Q, ops.to_device(Ub))                   # [m, k]
        return (-1 * U.t()).contiguous()                       # :30


PSD_EIG_MAX_M = 4096             # _make_cov_psd: largest Gram whose spectrum is checked (SOBER/_utils.py:122-124)


def _make_cov_psd(A, max_iter: int = 10):
    """``SafeTensorOperator.make_cov_psd`` (``SOBER/_utils.py:128-154``) for the Nystrom Gram, on the device.

    The reference tests exact symmetry + Cholesky + ``eig >= 0``; a kernel Gram computed in floating point
    is never bitwise symmetric, so its repair branch ``cov <- sqrt(cov * cov.T)`` always runs: that is done here
    unconditionally.  The follow-up PSD test is the reference's: Cholesky AND no negative eigenvalue -- the spectrum
    from the symmetric solver (the matrix is exactly symmetric after the repair; the reference's general ``eig`` sees
    the same eigenvalues up to round-off) for Grams of up to ``PSD_EIG_MAX_M`` points.  Beyond that (where the
    reference's own ``eig`` of an [m, m] matrix takes tens of minutes) Cholesky alone decides -- the one stated fork.
    When the test fails, the reference's diagonal-jitter loop is reproduced.
    """
    A = torch.sqrt(torch.nan_to_num(A) * torch.nan_to_num(A).T)

    def psd(M_):
        if int(torch.linalg.cholesky_ex(M_).info.item()) != 0:
            return False
        if M_.shape[0] > PSD_EIG_MAX_M:
            return True
        return bool((torch.linalg.eigvalsh(M_) >= 0).all())

    if not psd(A):
        n = A.shape[0]
        jitter = torch.full((n,), 1e-5, dtype=A.dtype, device=A.device)
        it = 0
        while not psd(A):
            A.diagonal().add_(jitter)
            jitter = jitter * 2
            it += 1
            if it > max_iter:
                A = torch.diag(torch.diagonal(A))
                break
    return A


class RecombinationEngine:
    def __init__(self, ops, comm=None):
        self.ops = ops
        self.comm = comm or LocalComm()

    # ------------------------------------------------------------------------------------------------
    def run(self, pts_local, gid0: int, n_total: int, pts_nys, num_pts: int, kernel, trace: EngineTrace | None = None,
            variant: str = "basq", init_weights=None, objective=None):
        """One recombination batch (see ``_run``); the launch stream is looked up once for the whole batch."""
        pin = getattr(self.ops, "pin_stream", None)
        if pin is None:
            return self._run(pts_local, gid0, n_total, pts_nys, num_pts, kernel, trace, variant, init_weights, objective)
        pin()
        try:
            return self._run(pts_local, gid0, n_total, pts_nys, num_pts, kernel, trace, variant, init_weights, objective)
        finally:
            self.ops.unpin_stream()

    def _run(self, pts_local, gid0: int, n_total: int, pts_nys, num_pts: int, kernel, trace: EngineTrace | None = None,
             variant: str = "basq", init_weights=None, objective=None, _async_allowed: bool = True):
        """Recombine.  ``pts_local`` = this rank's contiguous slice ``[gid0, gid0 + len)`` of the pool.

        ``variant="basq"`` follows ``BASQ/_rchq.py`` (uniform start weights, ``init_weights`` ignored);
        ``variant="sober"`` follows ``SOBER/_rchq.py``: ``init_weights`` (this rank's slice of them)
        are honoured and zero-weight points dropped, the Nystrom Gram goes through ``make_cov_psd``, the ragged
        remainder is additionally added to sets ``0..N_rest-1`` (:127-135), and an elimination that finds no
        positive entry stops early instead of failing (:240-242).

        ``objective`` (sober only): ``-calc_obj(pts_rec)`` (``SOBER/_rchq.py:67-69``), one value per local candidate.
        The reference can only execute its objective branch when the pool fits a single reduction (``:77-104``); for
        larger pools it raises at ``:140-142`` -- and so does this engine.

        Returns ``(idx int64[<=num_pts] ascending, w float64)`` on the ops device (identical on every rank).
        """
        ops, comm = self.ops, self.comm
        if variant not in ("basq", "sober"):
            raise ValueError(variant)
        sober = variant == "sober"
        rng_state = torch.get_rng_state() if (ASYNC_ROUNDS and _async_allowed and trace is None) else None
        if n_total >= 2 ** 31:
            raise ValueError("pool sizes >= 2^31 are not supported")
        if n_total == 0:                                        # empty pool: nothing to select (the reference returns [])
            return (torch.empty(0, dtype=torch.int64, device=getattr(ops, "device", "cpu")),
                    torch.empty(0, dtype=torch.float64, device=getattr(ops, "device", "cpu")))
        pts_nys = ops.to_device(pts_nys, torch.float64)
        pts_local = ops.to_device(pts_local, torch.float64)
        m, d = pts_nys.shape
        Rl = pts_local.shape[0]
        base, post, warp = kernel.base, kernel.posterior, kernel.warp
        # an opaque callable (the reference's own ``kernel`` contract): no packing, no fused kernel -- the candidates
        # stay raw [R, d] rows and every round's block sums come from dense chunks (``_opaque_message``)
        opaque = bool(getattr(kernel, "opaque", False))
        if opaque and sober:
            raise NotImplementedError("the SOBER variant needs a structured kernel (basq_amd.kernels)")
        if getattr(kernel, "jitter", 0.0) != 0.0:
            # wsabil/wsabim_kernel add `jitter` to entries [k][k] of every block (_wsabi.py:223,247), UNweighted by the
            # warped means; the reference hard-codes jitter = 0 (_wsabi.py:56) and the fused path carries no such term
            raise NotImplementedError("WsabiKernel.jitter != 0 is not supported by the fused recombination path")
        spec = None if opaque else base.spec(d)
        kp = d if opaque else ops.kp(d)
        kscale = 1.0 if opaque else spec.outputscale
        t_all = time.perf_counter()

        center = None if opaque else ops.col_mean(pts_nys)
        q = min(num_pts - 1, m)                                 # rank of svd_lowrank's output (reduced QR clips at m)
        s = q + 1
        S = 2 * s                                               # :50

        # ---- Nystrom-side operands of the block sums (no dependence on the basis) ----------------------
        t0 = time.perf_counter()
        nys_rows = [pts_nys]
        diag_noise, n_obs = 0.0, 0
        if post is not None:
            Xo = ops.to_device(post.Xobs, torch.float64)
            n_obs = Xo.shape[0]
            nys_rows.append(Xo)
            diag_noise = post.noise
        m_ext = m + n_obs
        q_ext = q
        wrow = 0
        if warp != "none" and diag_noise != 0.0:
            # an all-zero packed row has kernel value 1 with every candidate: its block sum is the
            # kernel-weighted set weight needed by the diagonal-noise term of wsabil_kernel
            zero_row_idx = m_ext
            m_ext += 1
            q_ext = q + 1
            wrow = q + 1
        nys_ext = None
        if not opaque:
            nys_cat = torch.cat(nys_rows, 0) if len(nys_rows) > 1 else pts_nys
            nys_ext = ops.pack(spec, nys_cat, center, ROLE_A, pad_rows_to=64)
            if nys_ext.shape[0] < ((m_ext + 63) // 64) * 64:
                nys_ext = torch.cat([nys_ext, ops.zeros(64, kp)], 0)
            if wrow:
                nys_ext[zero_row_idx].zero_()

        # ---- candidate state ---------------------------------------------------------------------------
        cand = (pts_local if Rl > 0 else ops.zeros(1, d)) if opaque else ops.pack(spec, pts_local, center, ROLE_B)
        mu, gid = ops.init_state(Rl, gid0, n_total)
        wx = None
        if warp != "none":
            wx = kernel.mean(ops, pts_local, center) if Rl > 0 else ops.empty(1)
        off, R = gid0, n_total
        obj_full = obj_live = None
        if objective is not None:
            if not sober:
                raise ValueError("an objective is part of the SOBER variant only")
            if comm.world > 1 or post is not None or warp != "none":
                raise NotImplementedError("objective row: single process, stationary kernels only")
            obj_full = obj_live = ops.to_device(objective, torch.float64).reshape(-1)
            if obj_full.shape[0] != Rl:
                raise ValueError("objective must have one entry per candidate")
        if sober and init_weights is not None:
            # SOBER/_rchq.py:60-64: start from the given weights, drop the zero-weight points up front
            w0 = ops.to_device(init_weights, torch.float64)
            if w0.shape[0] != Rl:
                raise ValueError("init_weights must have one entry per local candidate")
            nz = torch.nonzero(w0 != 0).reshape(-1)
            cand, mu, gid = cand[nz].contiguous(), w0[nz].contiguous(), gid[:Rl][nz].contiguous()
            if obj_live is not None:
                obj_live = obj_live[nz].contiguous()
            if wx is not None:
                wx = wx[nz].contiguous()
            Rl = int(nz.numel())
            counts = torch.tensor([float(Rl)], dtype=torch.float64, device=mu.device)
            if comm.world > 1:
                counts = comm.all_gather(counts).reshape(-1)
            counts = [int(v) for v in counts.cpu()]
            off, R = sum(counts[:comm.rank]), sum(counts)
            if Rl == 0:                                         # keep pointers valid for empty shards
                cand, mu, gid = ops.zeros(1, kp), ops.zeros(1), ops.zeros(1, dtype=torch.int64)

        # ---- round-1 block sums are queued BEFORE the basis: they do not depend on U, and the host's RNG draw
        #      for the range finder then overlaps with the largest kernel of the batch -----------------------
        pre = None
        late = None                                             # deferred part of the round-1 block sums
        cls = None                                              # inherited class MESSAGES: dict(M [C + 1, rows, S], C, reg_blocks)
        use_classes = CLASS_SUMS and not opaque and not sober and warp != "wsabim"

        def timed_blocksum(p_lo, p_hi, geo_, S_, n_ch, out, class_mod=0, class0=0):
            """One block-sum launch over the local positions [p_lo, p_hi) (+ HIP events for the roofline line)."""
            ev0 = ops.record_event() if (trace is not None and trace.time_kernels) else None
            ops.blocksum(spec, nys_ext, m_ext, cand[p_lo:], mu[p_lo:], None if wx is None else wx[p_lo:], p_hi - p_lo,
                         off + p_lo, geo_.n_full, S_, n_ch, out=out, class_mod=class_mod, class0=class0)
            if ev0 is not None and p_hi > p_lo:
                # pairs this launch evaluates: one class launch covers n_ch of class_mod classes of its range
                frac = (n_ch / class_mod) if class_mod else 1.0
                trace.kernel_events.append((ev0, ops.record_event(), dict(pairs=float(p_hi - p_lo) * m_ext * frac,
                                                                         R=(p_hi - p_lo) * frac, m=m_ext, S=S_, chunks=n_ch)))

        def irregular_block_sums(geo_, S_, reg_blocks):
            """Block sums of the candidates the class partials do not cover (global positions >= reg_blocks * S: further
            blocks + the ragged tail), one chunk -> ``(Xirr [1, m_ext, S], totirr [1, S])``."""
            Xirr, totirr = ops.empty(1, m_ext, S_), ops.empty(1, S_)
            reg_hi = min(max(reg_blocks * S_ - off, 0), Rl)              # local end of the regular region
            timed_blocksum(reg_hi, Rl, geo_, S_, 1, (Xirr, totirr))
            return Xirr, totirr

        def evaluate_block_sums(geo_, S_, defer_last=False):
            """A fresh evaluation of one round's block sums -> ``(Xbuf [n, m_ext, S], totbuf [n, S], n, C, reg_blocks, late_fn)``.

            C >= 2: the regular region -- the first ``reg_blocks`` (a multiple of C) blocks -- is summed per residue
            class (slots 0..C-1), the rest (further blocks + ragged tail) is one contiguous chunk (slot C = n - 1).
            C == 1 (small rounds, variants without class sums): plain contiguous chunks.  ``defer_last``: the last chunk
            / class (and the irregular chunk) are returned as ``late_fn`` instead of being launched (round 1: they run
            behind the range finder's GPU work)."""
            C = _classes_for(geo_.nb, m_ext, S_, kp // 4) if (use_classes and S_ == S) else 1
            if C == 1:
                n_ch = choose_chunks(local_blocks(off, Rl, geo_), m_ext, S_, kp // 4)
                Xbuf, totbuf = ops.empty(n_ch, m_ext, S_), ops.empty(n_ch, S_)
                p_split = _late_split(off, Rl, geo_.n_full, S_, n_ch, LATE_CHUNKS) if (defer_last and Rl > 0) else None
                if p_split is None:
                    timed_blocksum(0, Rl, geo_, S_, n_ch, (Xbuf, totbuf))
                    return Xbuf, totbuf, n_ch, 1, 0, None
                # the last chunk(s) are launched behind the range finder's GPU work; same chunk boundaries, same sums
                timed_blocksum(0, p_split, geo_, S_, n_ch - LATE_CHUNKS, (Xbuf[:n_ch - LATE_CHUNKS], totbuf[:n_ch - LATE_CHUNKS]))
                return (Xbuf, totbuf, n_ch, 1, 0,
                        lambda: timed_blocksum(p_split, Rl, geo_, S_, LATE_CHUNKS, (Xbuf[n_ch - LATE_CHUNKS:], totbuf[n_ch - LATE_CHUNKS:])))
            reg_blocks = (geo_.nb // C) * C
            Xbuf, totbuf = ops.empty(C + 1, m_ext, S_), ops.empty(C + 1, S_)
            reg_hi = min(max(reg_blocks * S_ - off, 0), Rl)              # local end of the regular region
            irregular = lambda: timed_blocksum(reg_hi, Rl, geo_, S_, 1, (Xbuf[C:C + 1], totbuf[C:C + 1]))   # noqa: E731
            if defer_last:
                L = max(1, min(LATE_CLASSES, C - 1))             # classes evaluated behind the range finder's GPU work
                timed_blocksum(0, reg_hi, geo_, S_, C - L, (Xbuf[:C - L], totbuf[:C - L]), class_mod=C, class0=0)

                def late_fn():
                    timed_blocksum(0, reg_hi, geo_, S_, L, (Xbuf[C - L:C], totbuf[C - L:C]), class_mod=C, class0=C - L)
                    irregular()

                return Xbuf, totbuf, C + 1, C, reg_blocks, late_fn
            timed_blocksum(0, reg_hi, geo_, S_, C, (Xbuf[:C], totbuf[:C]), class_mod=C, class0=0)
            irregular()
            return Xbuf, totbuf, C + 1, C, reg_blocks, None

        if R > S and not opaque:
            geo = RoundGeometry.of(R, S)
            pre = evaluate_block_sums(geo, S, defer_last=True)
            late = pre[5]

        # ---- Nystrom basis (one Gaussian draw on rank 0, as in the reference) ----------------------------
        # One rank: the range finder goes to a second stream.  Its ~2 ms of latency-bound steps (panel Cholesky,
        # triangular solves, Box-Muller, the host's k x k SVD) then hide under the round-1 block sums instead of
        # following them; the throughput-bound GEMMs simply share the chip.
        import contextlib

        two_streams = (BASIS_SIDE_STREAM and comm.world == 1 and hasattr(ops, "side_stream")
                       and (trace is None or not (trace.host_sync or trace.time_kernels)))   # timed kernels run alone
        if two_streams and late is not None:
            late()                                              # nothing is deferred: everything overlaps anyway
            late = None
        basis_ctx = ops.side_stream() if two_streams else contextlib.nullcontext()
        with basis_ctx as main_stream:
            U = self._basis(ops, comm, kernel, pts_nys, center, m, q, num_pts, sober, opaque, late, trace)
            if two_streams:
                U.record_stream(main_stream)
        if trace is not None:
            if trace.host_sync:
                ops.synchronize()
            trace.add_time("basis", time.perf_counter() - t0)
            if trace.keep_tensors:
                trace.U = U.clone()

        # ---- extended contraction matrix: posterior correction / warping folded in by linearity --------
        Um = U
        if warp != "none":
            mu_pt = kernel.mean(ops, pts_nys, center)
            Um = (U * mu_pt.unsqueeze(0)).contiguous()
        U_cols = [Um]
        if post is not None:
            W = ops.to_device(post.W, torch.float64)
            Bmat = base.dense(ops, pts_nys, Xo, center) @ W      # [m, n_obs] (small library GEMM, once per batch)
            U_cols.append(-(Um @ Bmat))
            if warp == "wsabim":
                # B^T, zero-padded to whole MFMA fragments: the A operand of the fused squared-covariance block sums
                bmatT = ops.zeros(((n_obs + 3) // 4) * 4, ((m + 63) // 64) * 64)
                bmatT[:n_obs, :m] = Bmat.t()
        U_ext = torch.cat(U_cols, 1) if len(U_cols) > 1 else Um
        if wrow:
            sel = ops.zeros(1, m_ext)
            U_ext = torch.cat([torch.cat([U_ext, ops.zeros(q, 1)], 1), sel], 0)
            U_ext[q, zero_row_idx] = 1.0 / kscale
        U_ext = U_ext.contiguous()
        diagU = Um if diag_noise != 0.0 else None
        if trace is not None:
            if trace.host_sync:
                ops.synchronize()
            trace.add_time("setup", time.perf_counter() - t0)

        # ---- rounds without a host round trip (one rank, plain block sums) ----------------------------------
        # The survivor count of a round depends on the data through two facts only (how many sets were kept, whether
        # the last set -- owner of the ragged tail -- is one of them), so the next round's geometry is a closed form a
        # one-thread kernel evaluates into a device-resident descriptor; every launch of the round reads its candidate
        # range from there.  The host enqueues all rounds that are CERTAINLY not the final one (lower bound of the
        # survivor count > S) without waiting, then reads the descriptor once and finishes round by round below.
        if (ASYNC_ROUNDS and trace is None and comm.world == 1 and not opaque and not sober and obj_full is None
                and warp != "wsabim" and hasattr(ops, "round_next") and R > S and _async_allowed):
            n_keep_exp = s                                       # a regular round keeps s = S/2 sets
            geo_t = ops.geo_init(64, R, S, (pre[4] * S) if (pre is not None and pre[3] >= 2) else 0)
            r = 0
            R_lo = R_up = R
            plan_C = None                                        # classes planned for a fresh evaluation (set below)
            while R_lo > S:
                g_row = geo_t[r]
                Mc, C_cur, parts = None, 1, None
                if cls is not None:                              # inside an epoch: regrouped class messages + the rest
                    Mc, C_cur = cls["M"], cls["C"]
                    Xirr, totirr = ops.empty(1, m_ext, S), ops.empty(1, S)
                    ops.blocksum_geo(spec, nys_ext, m_ext, cand, mu, wx, g_row, 2, S, 1, out=(Xirr, totirr))
                    ops.project_chunks(U_ext, q_ext, m_ext, Xirr, totirr, 1, S, kscale, out=Mc[C_cur:C_cur + 1])
                    parts = Mc
                else:
                    if pre is not None:                          # round 1: launched before the basis, host geometry
                        Xpart, totpart, n_chunks, C_cur = pre[:4]
                        pre = None
                    else:
                        C_cur = plan_C if plan_C is not None else 1
                        if C_cur >= 2:
                            n_chunks = C_cur + 1
                            Xpart, totpart = ops.empty(n_chunks, m_ext, S), ops.empty(n_chunks, S)
                            ops.blocksum_geo(spec, nys_ext, m_ext, cand, mu, wx, g_row, 1, S, C_cur,
                                             out=(Xpart[:C_cur], totpart[:C_cur]), class_mod=C_cur)
                            ops.blocksum_geo(spec, nys_ext, m_ext, cand, mu, wx, g_row, 2, S, 1,
                                             out=(Xpart[C_cur:], totpart[C_cur:]))
                        else:
                            n_chunks = choose_chunks(R_lo // S, m_ext, S, kp // 4)
                            Xpart, totpart = ops.blocksum_geo(spec, nys_ext, m_ext, cand, mu, wx, g_row, 3, S, n_chunks)
                    if C_cur >= 2:
                        Mc = ops.project_chunks(U_ext, q_ext, m_ext, Xpart, totpart, n_chunks, S, kscale)
                        parts = Mc
                    else:
                        parts = ops.project(U_ext, q_ext, m_ext, Xpart, totpart, n_chunks, S, kscale).unsqueeze(0)
                    del Xpart, totpart
                if diag_noise != 0.0:
                    # predictive_covariance's noise on the ragged tail block (entries [k][k], tail point k x Nystrom row
                    # k): one more message row carries the tail weights; its length is known on the device only, so the
                    # row is always there (all zeros without a tail: the extra terms vanish)
                    rows = parts.shape[1]
                    buf = ops.empty(1, rows + 1, S)
                    ops.sum_parts(parts, out=buf[0, :rows])
                    ops.tail_weights_geo(mu, wx, g_row, S, buf[0, rows])
                    XcarT, tot = ops.finalize(buf, 1, rows + 1, q, S, diagU, m, min(m, S), diag_noise, wrow, rows,
                                              min(m, S), geo_row=g_row)
                else:
                    XcarT, tot = ops.finalize(parts, parts.shape[0], parts.shape[1], q, S, None, m, min(m, S), 0.0, 0, 0, 0)
                PhiT = ops.nullspace(XcarT, s, S)
                keep_rank, kept, w_star, info = ops.car_eliminate(PhiT, tot.clone(), S, s)
                cls = None
                if Mc is not None and C_cur >= 2:
                    Mn = ops.empty(C_cur // 2 + 1, Mc.shape[1], S)
                    ops.regroup_classes(Mc[:C_cur], kept, w_star, tot, out=Mn[:C_cur // 2])
                    cls = dict(M=Mn, C=C_cur // 2, reg_blocks=None)
                # bounds of the next survivor count; the class plan of the next fresh evaluation follows the lower one
                R_lo_n = (R_lo // S) * n_keep_exp
                R_up_n = (R_up // S) * n_keep_exp + (S - 1)
                plan_C = None
                if cls is None:
                    plan_C = _classes_for(R_lo_n // S, m_ext, S, kp // 4) if use_classes else 1
                ops.round_next(g_row, info, keep_rank, S, -1 if cls is not None else (plan_C if plan_C >= 2 else 0),
                               True, geo_t[r + 1])
                cand, mu, gid, wx = ops.reweight_compact_geo(cand, mu, gid, wx, g_row, info, R_up, S, kp, keep_rank,
                                                             w_star, tot, R_up_n)
                r += 1
                R_lo, R_up = R_lo_n, R_up_n
            row = ops.to_host(geo_t[r], "geo_row").tolist()      # the ONE wait of the asynchronous rounds
            if row[3] != 0:
                # an elimination failed or did not keep half of the sets: regrouped messages may be invalid -- repeat
                # the batch with one read-back per round (same generator state, same results as that path alone)
                torch.set_rng_state(rng_state)
                return self._run(pts_local, gid0, n_total, pts_nys, num_pts, kernel, trace, variant, init_weights,
                                 objective, _async_allowed=False)
            R = Rl = int(row[0])
            if cls is not None:
                cls["reg_blocks"] = int(row[2]) // S
            elif plan_C is not None and plan_C >= 2:
                pass                                             # the next fresh evaluation re-plans from the exact R

        # ---- rounds -------------------------------------------------------------------------------------
        while True:
            if R <= s:                                           # :60-63 nothing to reduce
                gids, mus = self._gather_survivors(gid, mu, Rl, R, off, S)
                keep = mus > 0
                idx, w = gids[keep], mus[keep]
                break
            final = R <= S                                       # :65-74 single reduction of the points
            if obj_full is not None and not final:
                raise RuntimeError("recombination with an objective needs a pool of at most 2 * num_pts points: the "
                                   "reference fails here too (SOBER/_rchq.py:140-142 adds a [S, 1] sum in place to a "
                                   "[1, S] buffer)")
            S_r = R if final else S
            geo = RoundGeometry.of(R, S_r)
            t0 = time.perf_counter()
            C_cur, reg_blocks, msg, Mc = 1, 0, None, None
            # One rank, no extra message row: the class messages go to the finalize kernel as they are -- it adds its
            # parts in index order, which is exactly the sum a separate launch would have formed first.
            sum_classes_here = comm.world > 1 or (diag_noise != 0.0 and geo.n_tail > 0)
            if opaque:
                with _Timer(ops, trace, "blocksum"):
                    Xpart, totpart = self._opaque_block_sums(kernel, pts_nys, cand, mu, Rl, off, geo.n_full, S_r, m)
                n_chunks = 1
            elif cls is not None and not final and S_r == S:
                # inside an epoch: the class messages were regrouped from the previous round's; only the candidates they
                # do not cover are evaluated (a few blocks + the ragged tail)
                Mc, C_cur, reg_blocks = cls["M"], cls["C"], cls["reg_blocks"]
                with _Timer(ops, trace, "blocksum"):
                    Xirr, totirr = irregular_block_sums(geo, S_r, reg_blocks)
                with _Timer(ops, trace, "project"):
                    ops.project_chunks(U_ext, q_ext, m_ext, Xirr, totirr, 1, S_r, kscale, out=Mc[C_cur:C_cur + 1])
                    msg = ops.sum_parts(Mc) if sum_classes_here else Mc
            else:
                with _Timer(ops, trace, "blocksum"):
                    if pre is not None:
                        Xpart, totpart, n_chunks, C_cur, reg_blocks = pre[:5]
                        pre = None
                    else:
                        Xpart, totpart, n_chunks, C_cur, reg_blocks, _ = evaluate_block_sums(geo, S_r)
                if C_cur >= 2:
                    # start of an epoch: one message per residue class; the [m, S] partials are not needed again
                    with _Timer(ops, trace, "project"):
                        Mc = ops.project_chunks(U_ext, q_ext, m_ext, Xpart, totpart, n_chunks, S_r, kscale)
                        msg = ops.sum_parts(Mc) if sum_classes_here else Mc
                    del Xpart, totpart
            cls = None
            if msg is None and not opaque:                       # plain rounds: chunk partials -> one projection
                if sober and not final and geo.n_tail > 0:
                    # SOBER/_rchq.py:127-135: the remainder's kernel columns also go to sets 0..N_rest-1 (no weight added)
                    t0l = max(geo.n_full - off, 0)               # first local tail position
                    if t0l < Rl:
                        Xt, _ = ops.blocksum(spec, nys_ext, m_ext, cand[t0l:], mu[t0l:], None if wx is None else wx[t0l:],
                                             Rl - t0l, off + t0l - geo.n_full, S_r, S_r, 1)
                        Xpart = torch.cat([Xpart, Xt], 0)
                        totpart = torch.cat([totpart, ops.zeros(1, S_r)], 0)
                        n_chunks += 1
                with _Timer(ops, trace, "project"):
                    msg = ops.project(U_ext, q_ext, m_ext, Xpart, totpart, n_chunks, S_r, kscale)
            elif msg is None:
                with _Timer(ops, trace, "project"):
                    msg = ops.project(U_ext, q_ext, m_ext, Xpart, totpart, n_chunks, S_r, kscale)
            if warp == "wsabim":
                # + U @ (0.5 sum mu cov^2): the one term of wsabim_kernel that is not linear in the block sums
                with _Timer(ops, trace, "wsabim_sq"):
                    E = self._wsabim_square_term(spec, nys_ext, m, n_obs, bmatT, cand, mu, Rl, off, geo, S_r, diag_noise, kp)
                    msg[1:q + 1] += ops.matmul(U, E)
            tail_row, n_tail_diag = 0, 0
            if diag_noise != 0.0 and not final and geo.n_tail > 0:
                # the ragged tail is a kernel block of its own (:91-99): predictive_covariance adds the noise to ITS
                # entries [k][k] too (tail point k x Nystrom row k).  One more message row carries the tail weights.
                tailw = ops.zeros(S_r)
                t0l = max(geo.n_full - off, 0)                   # first local tail position
                if t0l < Rl:
                    k0 = off + t0l - geo.n_full
                    tailw[k0:k0 + (Rl - t0l)] = mu[t0l:Rl] if wx is None else mu[t0l:Rl] * wx[t0l:Rl]
                msg = torch.cat([msg, tailw.unsqueeze(0)], 0)
                tail_row, n_tail_diag = msg.shape[0] - 1, min(m, geo.n_tail)
            if trace is not None:
                if trace.host_sync:
                    ops.synchronize()
                trace.add_time("blocksum+project", time.perf_counter() - t0)
                t0 = time.perf_counter()
            if obj_full is not None:
                # SOBER/_rchq.py:78-104: one more feature per point, its objective (here still weighted by mu, like
                # every other message row); the reduction then keeps q + 2 points and the thinning removes one more
                idx, w = self._reduce_with_objective(msg, obj_live, obj_full, gid, mu, Rl, R, q, trace)
                break
            parts = comm.all_gather(msg) if comm.world > 1 else (msg if msg.dim() == 3 else msg.unsqueeze(0))
            M = S_r
            replicate = REPLICATED_REDUCTION and comm.world > 1
            if comm.rank == 0 or replicate:
                XcarT, tot = ops.finalize(parts, parts.shape[0], parts.shape[1], q, S_r, diagU, m, min(m, S_r), diag_noise, wrow,
                                          tail_row, n_tail_diag)
                if GPU_NULLSPACE:
                    with _Timer(ops, trace, "nullspace"):
                        PhiT = ops.nullspace(XcarT, s, M)        # :140-143 (rows = null-space vectors)
                else:
                    t1 = time.perf_counter()
                    Xh = ops.to_host(XcarT, "xcar")
                    with _lapack_threads(HOST_SVD_THREADS):
                        Vh = torch.linalg.svd(Xh)[2]             # :140 full SVD of [s, M] on host LAPACK
                    PhiT = ops.from_host(Vh[-(M - s):, :], "phit")
                    if trace is not None:
                        trace.add_time("host_svd", time.perf_counter() - t1)
                mu_car = tot.clone()
                with _Timer(ops, trace, "eliminate"):
                    keep_rank, kept, w_star, info = ops.car_eliminate(PhiT, mu_car, M, s)
            Mn = None
            if Mc is not None and C_cur >= 2 and not final and (comm.world == 1 or replicate):
                # Enqueued BEFORE the host waits for this round's outcome: if exactly half of the sets survive (checked
                # below), the next round's class messages are a gather + rescale of this round's; otherwise the result
                # is dropped (the kernel tolerates a short survivor list).
                Mn = ops.empty(C_cur // 2 + 1, Mc.shape[1], S_r)
                ops.regroup_classes(Mc[:C_cur], kept, w_star, tot, out=Mn[:C_cur // 2])
            if comm.world > 1 and not replicate:
                # one broadcast of the (tiny) reduction result: info | kept | w_star | keep_rank | tot
                res = ops.empty(2 + 4 * M)
                if comm.rank == 0:
                    res[0:2] = info.to(torch.float64)
                    res[2:2 + M] = kept.to(torch.float64)
                    res[2 + M:2 + 2 * M] = w_star
                    res[2 + 2 * M:2 + 3 * M] = keep_rank.to(torch.float64)
                    res[2 + 3 * M:] = tot
                comm.broadcast(res)
                head = ops.to_host(res[:2 + M], "head")
                hl = head.tolist()                               # one conversion (iterating a tensor costs ~1 us/element)
                n_keep, status = int(hl[0]), int(hl[1])
                kept_list = [int(v) for v in hl[2:2 + n_keep]]
                w_star = res[2 + M:2 + 2 * M].contiguous()
                keep_rank = res[2 + 2 * M:2 + 3 * M].to(torch.int32)
                tot = res[2 + 3 * M:].contiguous()
            else:
                ik_buf = getattr(info, "_base", None)            # HipOps hands out views of one [info | kept] buffer
                both = ik_buf if (ik_buf is not None and ik_buf.numel() == info.numel() + kept.numel()) \
                    else torch.cat([info, kept])
                head = ops.to_host(both, "head")                 # one D2H for status + survivor list
                hl = head.tolist()
                n_keep, status = hl[0], hl[1]
                kept_list = hl[2:2 + n_keep]
            if status != 0 and not sober:
                raise RuntimeError("Caratheodory elimination: a null vector has no positive entry "
                                   "(the reference fails here too: argmin of an empty tensor, _rchq.py:152)")
            if trace is not None:
                if trace.host_sync:
                    ops.synchronize()
                trace.add_time("reduce", time.perf_counter() - t0)
                rec = dict(R=R, S=S_r, nb=geo.nb, n_tail=geo.n_tail, kept=kept_list)
                if trace.keep_tensors:
                    rec["tot"] = tot.cpu()
                    if comm.rank == 0:
                        rec["XcarT"] = XcarT.cpu()
                    rec["w_star"] = w_star[:n_keep].cpu()
                trace.rounds.append(rec)
            if final:
                gids, _ = self._gather_survivors(gid, mu, Rl, R, off, S)
                kept_t = torch.tensor(kept_list, dtype=torch.int64, device=gids.device)
                idx, w = gids[kept_t], w_star[:n_keep].clone()   # :69-73
                break
            t0 = time.perf_counter()
            if Mc is not None and C_cur >= 2 and 2 * n_keep == S_r and status == 0:
                # exactly half of the sets survived: the next round's class messages are a gather + rescale of this round's
                if Mn is None:
                    Mn = ops.empty(C_cur // 2 + 1, Mc.shape[1], S_r)
                    ops.regroup_classes(Mc[:C_cur], res[2:2 + M].to(torch.int32), w_star, tot, out=Mn[:C_cur // 2])
                cls = dict(M=Mn, C=C_cur // 2, reg_blocks=reg_blocks // 2)
            new_off, new_Rl = next_shard(off, Rl, geo, kept_list)
            cand, mu, gid, wx = ops.reweight_compact(cand, mu, gid, wx, Rl, off, geo.n_full, S_r, kp, keep_rank, w_star,
                                                     tot, n_keep, new_off, new_Rl)
            R = survivors_before(R, geo, kept_list)
            off, Rl = new_off, new_Rl
            if trace is not None:
                if trace.host_sync:
                    ops.synchronize()
                trace.add_time("compact", time.perf_counter() - t0)
        if trace is not None:
            if trace.host_sync:
                ops.synchronize()
            trace.add_time("total", time.perf_counter() - t_all)
        return idx, w

    # ------------------------------------------------------------------------------------------------
    def _basis(self, ops, comm, kernel, pts_nys, center, m, q, num_pts, sober, opaque, late, trace):
        """Nystrom Gram + range finder -> ``U [q, m]`` (identical on every rank)."""
        if SHARDED_BASIS and comm.world > 1 and not sober and not opaque:
            # every rank builds its row block of the Gram matrix and takes part in the range finder (no broadcast of U)
            shards = initial_shards(m, comm.world)
            r0, mr = shards[comm.rank]
            with _Timer(ops, trace, "basis.gram"):
                A_rows = kernel.dense(ops, pts_nys[r0:r0 + mr].contiguous(), pts_nys, center, diag_offset=r0) \
                    if mr else ops.zeros(0, m)
            return nystrom_basis(ops, _ShardedProducts(ops, comm, A_rows, shards, m), num_pts - 1, trace, overlap=late)
        if comm.rank == 0:
            with _Timer(ops, trace, "basis.gram"):
                A = kernel.dense(ops, pts_nys, pts_nys, center)
                if sober:
                    A = _make_cov_psd(A)
            U = nystrom_basis(ops, A, num_pts - 1, trace, overlap=late)
            del A
            assert U.shape[0] == q
        else:
            U = ops.empty(q, m)
            _skip_test_matrix_draw(ops, m, num_pts - 1)         # keep this rank's global generator in step with rank 0
            if late is not None:
                late()                                          # runs while rank 0 finishes the basis
        if comm.world > 1:
            comm.broadcast(U)
        return U

    def _opaque_block_sums(self, kernel, pts_nys, cand, mu, Rl, off, n_full, S, m):
        """Block sums of one round for an opaque callable: ``X_for_nys`` and ``tot_weights`` of ``_rchq.py:79-99`` as
        ``(Xpart [1, m, S], totpart [1, S])``, same layout as ``basq_blocksum_f64`` with one chunk.

        Chunked mode: ``C = kernel(pts_nys, chunk)`` ([m, nc] float64 on the device, at most ``chunk_bytes``) per chunk
        of consecutive candidates, summed into the sets by ``basq_dense_blocksum_f64`` in position order (the set
        weights through the same kernel with an all-ones row).  ``block_exact`` mode: the reference's own calls, one
        ``kernel(pts_nys, block)`` per block of S points and one for the ragged tail (needed when the callable's value
        depends on the block it is asked for, e.g. ``predictive_covariance``'s per-block noise diagonal)."""
        ops = self.ops
        E, T = ops.zeros(m, S), ops.zeros(1, S)
        if Rl == 0:
            return E.unsqueeze(0), T
        if kernel.block_exact:
            if self.comm.world > 1:
                raise NotImplementedError("block_exact callables run on a single rank (blocks straddle shard borders)")
            nb = n_full // S
            for i in range(nb):                                   # _rchq.py:81-86, call for call
                lo = i * S
                Kb = kernel.dense(ops, pts_nys, cand[lo:lo + S])
                ops.dense_blocksum(Kb, mu[lo:lo + S], lo, n_full, S, 1.0, E)
            if Rl > n_full:                                       # :91-99 the remainder, one call
                Kt = kernel.dense(ops, pts_nys, cand[n_full:Rl])
                ops.dense_blocksum(Kt, mu[n_full:Rl], n_full, n_full, S, 1.0, E)
            ones = ops.zeros(1, Rl) + 1.0
            ops.dense_blocksum(ones, mu, 0, n_full, S, 1.0, T)
            return E.unsqueeze(0), T
        nc_max = max(S, min(Rl, kernel.chunk_bytes // (8 * m)))
        ones = ops.zeros(1, min(nc_max, Rl)) + 1.0
        for p0 in range(0, Rl, nc_max):
            nc = min(nc_max, Rl - p0)
            Kc = kernel.dense(ops, pts_nys, cand[p0:p0 + nc])
            ops.dense_blocksum(Kc, mu[p0:p0 + nc], off + p0, n_full, S, 1.0, E)
            ops.dense_blocksum(ones[:, :nc], mu[p0:p0 + nc], off + p0, n_full, S, 1.0, T)
        return E.unsqueeze(0), T

    def _reduce_with_objective(self, msg, obj_live, obj_full, gid, mu, Rl, R, q, trace):
        """Single reduction with an objective row (``SOBER/_rchq.py:77-111``), one process.

        ``msg`` = ``[tot ; U @ block sums]`` of the R points (one set each).  The Caratheodory step runs on
        ``[1 ; features ; objective]`` (q + 2 rows); then, among the kept points, the weights move along the null vector
        of ``[features ; 1]`` -- oriented so that the weighted objective does not decrease -- until one more reaches
        zero (``:87-104``).  That last step is k <= q + 2 numbers: host LAPACK, as in the reference.
        """
        ops = self.ops
        obj_row = (obj_live[:Rl] * mu[:Rl]).reshape(1, -1)
        parts = torch.cat([msg[:q + 1], obj_row], 0).unsqueeze(0).contiguous()
        XcarT, tot = ops.finalize(parts, 1, q + 2, q + 1, R, None, 0, 0, 0.0, 0)
        s_car = q + 2
        if R > s_car:
            PhiT = ops.nullspace(XcarT, s_car, R)
            _, kept, w_star, info = ops.car_eliminate(PhiT, tot.clone(), R, s_car)
            head = ops.to_host(torch.cat([info, kept]), "head")
            hl = head.tolist()
            n_keep = hl[0]
            kept_pos = torch.tensor(hl[2:2 + n_keep], dtype=torch.int64)
            w_host = ops.to_host(w_star[:n_keep], "wobj").clone()
        else:                                                    # nothing to eliminate (V[-0:] is the whole of V, :235)
            w_host = ops.to_host(tot, "wobj").clone()
            live = w_host > 0
            kept_pos = torch.arange(R, dtype=torch.int64)[live]
            w_host = w_host[live]
        F = XcarT[1:q + 1].cpu()[:, kept_pos]                     # features of the kept points, without the objective
        obj_p = obj_full.cpu()[kept_pos]                          # (sic) :89 indexes the objective by POSITION
        A = torch.cat([F, torch.ones(1, len(kept_pos), dtype=torch.float64)], 0)
        with _lapack_threads(HOST_SVD_THREADS):
            direction = torch.linalg.svd(A)[2][-1]
        if torch.dot(obj_p, direction) < 0:
            direction = -direction
        pos = direction > 0
        ratio = torch.zeros(len(w_host), dtype=torch.float64)
        ratio[pos] = w_host[pos] / direction[pos]
        hit = torch.arange(len(w_host))[pos][torch.argmin(ratio[pos])]
        w_host = w_host - ratio[hit] * direction
        w_host[hit] = 0.0
        sel = w_host > 0
        kept_pos, w_host = kept_pos[sel], w_host[sel]
        if trace is not None:
            trace.rounds.append(dict(R=R, S=R, nb=1, n_tail=0, kept=[int(v) for v in kept_pos]))
        gids = gid[:Rl]
        return gids[kept_pos.to(gids.device)], ops.to_device(w_host)

    def _wsabim_square_term(self, spec, nys_ext, m, n_obs, bmatT, cand, mu, Rl, off, geo, S, diag_noise, kp):
        """E[j, s] = 0.5 * sum_{p in set s} mu_p * cov(pt_j, x_p)^2  with cov = k - K(pt,X) W K(X, x)  (_wsabi.py:240).

        ``cov`` is ``predictive_covariance``, which carries the likelihood noise on entry [k][k] of every block the
        reference builds: candidate p of a full block meets Nystrom row ``p mod S``, tail point k meets row k.

        Fused: one Gram launch for ``K(X, x_p)`` of the live candidates ([n_obs, Rl], the only per-candidate array),
        then ``basq_blocksum_sq_f64`` evaluates k, subtracts the correction (a second MFMA chain over the observations),
        squares and accumulates in registers -- no [m, candidates] covariance block exists.
        """
        ops = self.ops
        if Rl == 0:
            return ops.zeros(m, S)
        n4 = bmatT.shape[0]
        kobs = ops.zeros(n4, Rl) if n4 != n_obs else ops.empty(n4, Rl)
        ops.gram_into(spec, nys_ext[m:m + n_obs], n_obs, cand, Rl, kobs)      # rows m.. of nys_ext = packed observations
        n_ch = choose_chunks(local_blocks(off, Rl, geo), m, S, kp // 4)
        Epart = ops.blocksum_sq(spec, nys_ext, m, cand, mu, Rl, off, geo.n_full, S, n_ch, bmatT, kobs, n_obs, diag_noise)
        return Epart[0] if n_ch == 1 else Epart.sum(0)

    # ------------------------------------------------------------------------------------------------
    def _gather_survivors(self, gid, mu, Rl, R, off, cap):
        """All ranks' (gid, mu) of the R <= cap survivors, in global position order, on every rank."""
        comm, ops = self.comm, self.ops
        if comm.world == 1:
            return gid[:Rl], mu[:Rl]
        buf = ops.zeros(2 * cap + 2)
        buf[0] = float(off)
        buf[1] = float(Rl)
        buf[2:2 + Rl] = gid[:Rl].to(torch.float64)               # ids < 2^31: exact in float64
        buf[2 + cap:2 + cap + Rl] = mu[:Rl]
        allb = comm.all_gather(buf).cpu()
        gids = torch.empty(R, dtype=torch.int64)
        mus = torch.empty(R, dtype=torch.float64)
        for r in range(comm.world):
            o, n = int(allb[r, 0]), int(allb[r, 1])
            gids[o:o + n] = allb[r, 2:2 + n].to(torch.int64)
            mus[o:o + n] = allb[r, 2 + cap:2 + cap + n]
        return ops.to_device(gids), ops.to_device(mus)
