"""The Nystrom basis of ``ker_svd_sparsify`` (``BASQ/_rchq.py:28-31``): ``-svd_lowrank(kernel(pt, pt), q)[0].T``.

Restates ``torch._lowrank.get_approximate_basis`` / ``_svd_lowrank`` (torch 2.10, niter=2, square A so no
transposition).  What runs where:

* the Gaussian test matrix is drawn exactly where the reference draws it -- one ``torch.randn(m, q)`` from the CPU
  global generator -- as uniforms on the host + Box-Muller on the GPU (``basq_box_muller_f64``);
* the six ``[m, m] x [m, q]`` products on ``basq_skinny_gemm_f64`` (row-sharded over the ranks of a multi-GPU run);
* CholeskyQR (``basq_chol_factor_f64`` + ``basq_trsm_rows_f64``) instead of the five Householder QRs, an LQ reduction +
  ONE ``q x q`` SVD on host LAPACK instead of the ``[q, m]`` SVD.

The functions that wait for the host are generators: they ``yield`` the event they would block on, so that the engine can
interleave several batches from one host thread (``RecombinationEngine.run_many``); ``nystrom_basis`` is the blocking
form.
"""
from __future__ import annotations

import time

import torch

from . import _config as cfg


class _lapack_threads:
    def __init__(self, n=None):
        self.n = n or cfg.HOST_LAPACK_THREADS

    def __enter__(self):
        self.prev = torch.get_num_threads()
        if self.prev > self.n:
            torch.set_num_threads(self.n)
        return self

    def __exit__(self, *exc):
        if torch.get_num_threads() != self.prev:
            torch.set_num_threads(self.prev)
        return False


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


def wait_for(ev):
    """Block until ``ev`` has fired.  ``_config.SPIN_WAIT``: poll it instead of the runtime's blocking wait (whose wake-up is
    tens of microseconds: a synchronous batch waits 5-6 times, each with the GPU idle)."""
    if cfg.SPIN_WAIT:
        q = getattr(ev, "query", None)
        if q is not None:
            while not q():
                pass
            return
    ev.synchronize()


def drive(gen):
    """Run a step generator to completion, blocking on every event it yields -> its return value."""
    try:
        while True:
            wait_for(next(gen))
    except StopIteration as stop:
        return stop.value


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


def _mm_splitk(ops, A, B, want: int = 8):
    """``A @ B`` for a long contraction with a skinny output (<= 208 columns): the hand-written tall-skinny MFMA kernel
    (``basq_skinny_gemm_f64``).  Wider outputs, other dtypes and the CPU stand-in take the library path below: ONE
    batched library GEMM over K slices.

    rocBLAS tiles the output only (no split-K): ``[1e4,1e4] @ [1e4,99]`` is 79 work-groups on 256 CUs and
    ``[99,1e4] @ [1e4,99]`` a single one.  Viewing the K dimension as (splits, K/splits) -- strided views, no
    copies -- runs ``splits`` times more work-groups concurrently; the partial products are added in slice order.
    """
    sk = getattr(ops, "skinny_gemm", None) if cfg.OWN_RANGE_GEMM else None
    if sk is not None and A.dim() == 2 and B.dim() == 2 and B.shape[1] <= ops.SKINNY_MAX_N and B.stride(1) == 1 \
            and A.dtype == torch.float64 and B.dtype == torch.float64:
        # the hand-written tall-skinny MFMA kernel (basq_skinny_gemm_f64): A read once, split-K inside
        if A.stride(1) == 1 and A.stride(0) >= A.shape[1]:
            return sk(A, B, False)
        if A.stride(0) == 1 and A.stride(1) >= A.shape[0]:
            return sk(A.t(), B, True)                          # A is a transposed view: read the stored matrix
    M, K = A.shape
    c = _splits_for(K, want)
    if c == 1 or getattr(ops, "name", "") != "hip":
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
        if cfg.FUSED_CHOLQR and k <= getattr(ops, "CHOLQR_FUSED_MAX_Q", 0):
            X, info = ops.cholqr(G, X)                         # factor and solve in one launch, pipelined by panels
            flags.append(info)
        elif k <= getattr(ops, "CHOL_FACTOR_MAX_Q", 0):
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
        h2d = ops.from_pinned_side if (cfg.RAND_COPY_STREAM and hasattr(ops, "from_pinned_side")) else ops.from_pinned
        R = ops.box_muller(h2d(u), None if ut is None else ops.from_pinned(ut))
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

    def full(self):
        return self.A

    def qr_of_product(self, Q, flags, passes, transpose=False):
        """Orthonormal basis of ``range(A Q)`` (``transpose``: ``A^T Q``) by CholeskyQR -> ``[m, k]``."""
        return _cholqr(self.ops, self.at(Q) if transpose else self.a(Q), flags, passes)

    def lq_factors(self, Q):
        """``B = Q^T A = L1 L2 Qb^T`` by CholeskyQR2 on the rows of ``Y = A^T Q`` -> ``(G1, G2, i1, i2, k)`` with the factors in
        the lower triangles of ``G1``, ``G2`` (``nystrom_basis_steps``)."""
        ops = self.ops
        Y = self.at(Q)
        k = Y.shape[1]
        G1 = _mm_splitk(ops, Y.t(), Y, 32)                     # = B B^T
        if cfg.FUSED_CHOLQR and k <= getattr(ops, "CHOLQR_FUSED_MAX_Q", 0):
            Yq, i1 = ops.cholqr(G1, Y)                          # = (L1^-1 B)^T, G1 -> L1 in place
            G2 = _mm_splitk(ops, Yq.t(), Yq, 32)
            i2 = ops.chol_factor(G2)
        elif k <= getattr(ops, "CHOL_FACTOR_MAX_Q", 0):
            i1 = ops.chol_factor(G1)
            Yq = ops.trsm_rows(Y, G1)                           # = (L1^-1 B)^T
            G2 = _mm_splitk(ops, Yq.t(), Yq, 32)
            i2 = ops.chol_factor(G2)
        else:
            W1, i1 = ops.chol_inv(G1)
            Yq = _mm_splitk(ops, Y, W1, 1)
            G2 = _mm_splitk(ops, Yq.t(), Yq, 32)
            _, i2 = ops.chol_inv(G2)
        return G1, G2, i1, i2, k


class _ShardedProducts:
    """Rows ``[r0, r0 + mr)`` of ``A`` on this rank (multi-GPU, SURVEY 8e: the range finder no longer idles W-1 GPUs).

    ``A`` is a kernel Gram matrix -- symmetric up to the rounding of its entries -- so ``A^T Q`` and ``(Q^T A)^T`` are
    computed as ``A Q`` as well: every product is ``A_rows @ Q`` on each rank followed by ONE all-gather of the
    ``[mr, q]`` blocks (1 MB per rank at the headline size), after which all ranks hold the same ``[m, q]`` matrix and
    run the small replicated steps (the q x q Cholesky factors) identically.  Only rank 0 consumes the RNG: the
    Gaussian test matrix is broadcast.  The deviation from the single-GPU arithmetic (A for A^T) is at rounding
    level, far inside the stability margin of the selection (SURVEY finding 3); the gloo tests pin the indices.
    """

    def __init__(self, ops, comm, A_rows, shards, m, drawer=0, lockstep=True):
        """``drawer``: the rank that draws the Gaussian test matrix (broadcast from there).  ``lockstep``: the other ranks
        advance their CPU generators by the same amount (the default: a loop that samples from the global generator on every
        rank keeps seeing ONE stream).  ``run_many`` turns it off when every job carries its own seed -- every rank is then
        re-seeded before every batch, ANY rank draws the same matrix, and the 2.2-ms draw of a headline batch is dealt
        round-robin with the batch's owner instead of being repeated by all ranks (the generators are brought back in step once,
        after the last job)."""
        self.ops, self.comm, self.rows, self.shards, self.m = ops, comm, A_rows, shards, m
        self.mb = max(n for _, n in shards)
        self.drawer, self.lockstep = int(drawer) % comm.world, bool(lockstep)

    def draw(self, q, trace):
        if self.comm.rank == self.drawer:
            R = _gaussian_test_matrix(self.ops, self.m, q, trace).contiguous()
        else:
            if self.lockstep:
                _skip_test_matrix_draw(self.ops, self.m, q)     # same generator consumption on every rank
            R = self.ops.empty(self.m, q)
        return self.comm.broadcast(R, src=self.drawer)

    def a(self, Q):
        mr = self.rows.shape[0]
        blk = self.ops.zeros(self.mb, Q.shape[1])
        if mr:
            blk[:mr] = _mm_splitk(self.ops, self.rows, Q, 64)
        g = self.comm.all_gather(blk)                            # [W, mb, q]
        return torch.cat([g[r, :n] for r, (_, n) in enumerate(self.shards)], 0)

    at = a

    def full(self):
        blk = self.ops.zeros(self.mb, self.m)
        blk[:self.rows.shape[0]] = self.rows
        g = self.comm.all_gather(blk)
        return torch.cat([g[r, :n] for r, (_, n) in enumerate(self.shards)], 0)

    # -- CholeskyQR with the rows divided over the ranks (round 4) ---------------------------------------------------
    # ``X = A Q`` is born row-sharded (``X_r = A_rows @ Q``), and CholeskyQR needs X only through ``G = X^T X = sum_r X_r^T X_r``:
    # every rank multiplies and solves ITS rows -- ``G_r`` (q x q) all-gathered and added in rank order, the q x q Cholesky
    # replicated (it is the serial part), ``Q_r = X_r L^-T`` local -- and the blocks that travel are those of Q, not of X.
    # One small collective more per pass (80 KB per rank at q = 99); the Gram product and the triangular solve, which
    # every rank used to run on all m rows, divide by the rank count.
    def _gram_sum(self, Xr):
        ops, k = self.ops, Xr.shape[1]
        Gr = _mm_splitk(ops, Xr.t(), Xr, 32) if Xr.shape[0] > 0 else ops.zeros(k, k)
        return ops.sum_parts(self.comm.all_gather(Gr))          # rank order: the same G, bit for bit, on every rank

    def _gather_rows(self, Xr):
        blk = self.ops.zeros(self.mb, Xr.shape[1])
        if Xr.shape[0]:
            blk[:Xr.shape[0]] = Xr
        g = self.comm.all_gather(blk)
        return torch.cat([g[r, :n] for r, (_, n) in enumerate(self.shards)], 0)

    def _local_product(self, Q):
        return _mm_splitk(self.ops, self.rows, Q, 64) if self.rows.shape[0] else self.ops.zeros(0, Q.shape[1])

    def _sharded_ok(self, k):
        return cfg.SHARDED_CHOLQR and k <= getattr(self.ops, "CHOL_FACTOR_MAX_Q", 0)

    def _cholqr_rows(self, Xr, flags, passes):
        ops = self.ops
        for _ in range(passes):
            G = self._gram_sum(Xr)
            if Xr.shape[0] and cfg.FUSED_CHOLQR and Xr.shape[1] <= getattr(ops, "CHOLQR_FUSED_MAX_Q", 0):
                Xr, info = ops.cholqr(G, Xr)
                flags.append(info)
            else:
                flags.append(ops.chol_factor(G))
                Xr = ops.trsm_rows(Xr, G) if Xr.shape[0] else Xr
        return Xr, G

    def qr_of_product(self, Q, flags, passes, transpose=False):
        if not self._sharded_ok(Q.shape[1]):
            return _cholqr(self.ops, self.a(Q), flags, passes)
        Xr, _ = self._cholqr_rows(self._local_product(Q), flags, passes)
        return self._gather_rows(Xr)

    def lq_factors(self, Q):
        k = Q.shape[1]
        if not self._sharded_ok(k):
            return _DenseProducts.lq_factors(self, Q)
        flags = []
        Yq, G1 = self._cholqr_rows(self._local_product(Q), flags, 1)     # G1 -> L1 in place (every rank: the same bits)
        G2 = self._gram_sum(Yq)
        i2 = self.ops.chol_factor(G2)
        return G1, G2, flags[0], i2, k


class BasisResult:
    """What the range finder hands to the batch: ``U [min(q, m), m]`` and -- when the basis was formed without waiting for the
    host -- the pivot flag of its CholeskyQR passes (``bad``: a device int32 scalar, nonzero = a numerically rank-deficient
    panel) with ``fallback()``, which recomputes the basis by host Householder QR from the same Gaussian draw.  ``bad is None``:
    the basis is final."""

    def __init__(self, U, bad=None, fallback=None):
        self.U, self.bad, self.fallback = U, bad, fallback


def _host_range_finder(ops, prod, R, trace=None):
    """``torch.svd_lowrank``'s own steps with host Householder QR (the rare path: a CholeskyQR pivot flagged a numerically
    rank-deficient panel, or q exceeds m) -> ``U [k, m]``."""
    A = prod.full()                                            # (sharded: gathered)
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


FALLBACK_NOTE = ("range finder: a Cholesky pivot flagged a numerically rank-deficient panel (cond > ~1e6); the basis "
                 "was recomputed with host Householder QR from the same Gaussian draw")


def nystrom_basis_steps(ops, A, q_req: int, trace=None, overlap=None, notes=None):
    """``ker_svd_sparsify`` as a step generator -> :class:`BasisResult` (see :func:`nystrom_basis`).

    What the reference's result depends on is only (i) the Gaussian draw and (ii) the *range* of each intermediate
    ``Q``.  Two invariances of the recombination make the rest free:

    * row-sign flips of ``U`` change nothing, bit for bit, and more generally (round 4) ANY orthogonal rotation of the rows of
      ``U`` changes nothing but rounding: it turns every round's matrix ``X = [1 ; features]`` into ``diag(1, R) X`` -- same
      ``X^T X``, same first row -- so the Golub-Kahan right vectors, i.e. LAPACK's right Householder reflectors whose trailing
      rows are the null-space basis the elimination pivots on, are the same (``dlarfg`` maps ``(alpha, x)`` and ``(-alpha, -x)`` to
      the same ``tau, v``: even the signs uniqueness leaves open do not matter).  ``tests/test_oracle.py::
      test_selection_invariant_under_basis_rotations`` pins this on the reference's own op sequence;
    * so the five Householder QRs (host LAPACK in the reference) become CholeskyQR on the GPU, and the final ``[q, m]`` SVD of
      ``torch.svd_lowrank`` -- whose only effect on ``U`` is such a rotation of the orthonormal basis ``Q`` of the range -- is
      SKIPPED: ``U = -Q^T``.  That removes the sixth ``[m, m] x [m, q]`` product (``B^T = A^T Q``), its LQ reduction, the q x q host
      SVD and the batch's first host wait (``_config.BASIS_SVD = True`` restores them: round 3's path).

    Without a host wait the pivot flags of the CholeskyQR passes cannot be looked at here: they travel with the result
    (``BasisResult.bad``) and the batch checks them at its first read-back; a flagged basis (cond > ~1e6 panel) is recomputed by
    ``fallback()`` -- host Householder QR + the reference's SVD, from the same Gaussian draw.

    ``A``: the Gram matrix (a tensor) or a products object (``_DenseProducts`` / ``_ShardedProducts``).
    ``overlap``: optional callable that enqueues independent GPU work; it is called once, behind the range finder's launches.
    """
    prod = _DenseProducts(ops, A) if torch.is_tensor(A) else A
    m = prod.m
    with _Timer(ops, trace, "basis.randn"):
        R = prod.draw(q_req, trace)

    def fallback():
        if trace is not None:
            trace.timers["basis.fallback"] = trace.timers.get("basis.fallback", 0) + 1
        if notes is not None:
            notes.append(FALLBACK_NOTE)
        return _host_range_finder(ops, prod, R, trace)

    if cfg.GPU_RANGE_FINDER and q_req <= m:
        with _Timer(ops, trace, "basis.gpu_range"):
            flags = []
            Q = prod.qr_of_product(R, flags, 1)
            Q = prod.qr_of_product(Q, flags, 1, transpose=True)
            Q = prod.qr_of_product(Q, flags, 1)
            Q = prod.qr_of_product(Q, flags, 1, transpose=True)
            Q = prod.qr_of_product(Q, flags, 2)                  # the basis that is actually used
            if not cfg.BASIS_SVD:
                bad = torch.stack([f.reshape(()) for f in flags]).max().reshape(1)
                U = (-1 * Q.t()).contiguous()                    # :30 (the rotation by the SVD's left factor is immaterial)
                if overlap is not None:
                    overlap()
                return BasisResult(U, bad, fallback)
            # round 3's path: LQ of B = Q^T A ([k, m]) by CholeskyQR2 on its rows, formed on Y = B^T = A^T Q ([m, k]: tall,
            # row-parallel):  B = L1 L2 Qb^T  ->  the left singular vectors of B are those of L = L1 L2
            G1, G2, i1, i2, k = prod.lq_factors(Q)
            L = _mm_splitk(ops, torch.tril(G1), torch.tril(G2), 1)
            bad = torch.stack([f.reshape(()) for f in flags + [i1, i2]]).max()
            both, ready = ops.to_host_async(torch.cat([L.reshape(-1), bad.to(torch.float64).reshape(1)]), "basisL")
        if overlap is not None:
            overlap()
            overlap = None
        yield ready                                            # the ONE wait of the whole range finder
        with _Timer(ops, trace, "basis.host_svd", sync=False):
            Lh = both[:k * k].reshape(k, k)
            ok = int(both[k * k].item()) == 0
            if ok:
                with _lapack_threads(cfg.HOST_SVD_THREADS):
                    Ub = torch.linalg.svd(Lh)[0]
        if ok:
            with _Timer(ops, trace, "basis.gemm"):
                U = _mm_splitk(ops, Q, ops.to_device(Ub), 1)   # [m, k]
                return BasisResult((-1 * U.t()).contiguous())  # :30
    if overlap is not None:
        overlap()
    return BasisResult(fallback() if (cfg.GPU_RANGE_FINDER and q_req <= m) else _host_range_finder(ops, prod, R, trace))


def nystrom_basis(ops, A, q_req: int, trace=None, overlap=None):
    """Blocking form of :func:`nystrom_basis_steps` -> ``U`` (the pivot flag is read back and honoured here)."""
    res = drive(nystrom_basis_steps(ops, A, q_req, trace, overlap))
    if res.bad is not None and int(res.bad.cpu()[0]) != 0:
        return res.fallback()
    return res.U


def make_cov_psd(A, max_iter: int = 10):
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
        # Screening (round 4; the spectrum of a 500 x 500 Gram costs 10.8 ms on the GPU, 0.4 s on the host, against 6.7 ms for the
        # whole batch: profiles/r06_c_sober_phases.txt): if M - tau I has a Cholesky factor, every eigenvalue of M is >= tau,
        # three orders above what a computed spectrum can be off by (~1e-13 trace) -- both of the reference's tests pass,
        # and neither needs to be run.  Only a matrix that fails the screening takes the reference's sequence below.
        n = M_.shape[0]
        if n > 0:
            tau = (1e-10 * n ** 0.5) * torch.diagonal(M_).mean()             # (a device scalar: no read-back for it)
            shifted = M_.clone()
            shifted.diagonal().sub_(tau)
            if int(torch.linalg.cholesky_ex(shifted).info.item()) == 0:
                return True
        if int(torch.linalg.cholesky_ex(M_).info.item()) != 0:
            return False
        if M_.shape[0] > cfg.PSD_EIG_MAX_M:
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
