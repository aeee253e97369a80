"""One recombination batch in flight: state + the step generator the engine drives.

Mirrors ``BASQ/_rchq.py``: ``rc_kernel_svd`` (:34-40) = Nystrom basis + ``Mod_Tchernychova_Lyons`` (:43-130) whose
per-round reduction is ``Tchernychova_Lyons_CAR`` (:133-175).  What runs where:

=======================================  ==========================================================
step (reference lines)                   here
=======================================  ==========================================================
Gram + ``torch.svd_lowrank`` (:29)       ``_basis.nystrom_basis_steps`` (own GEMMs, CholeskyQR; no SVD: U = -Q^T)
hot loop + tail + tot (:79-99)           a *block-sum strategy* (below) -> the round's ``(q+1) x S`` message
round geometry (:76-78, :107-130)        closed form: on the host (``_partition``) or on the device
                                         (``basq_round_next_i64`` + the ``*_geo`` entries: no host wait per round)
``U_svd @ X_for_nys`` (:88)              ``basq_project_f64`` / ``basq_project_chunks_f64`` (f64 MFMA)
divide, ones column (:101, :138)         ``basq_finalize_f64``
full SVD -> null space (:140-143)        ``basq_nullspace_f64`` (the right reflectors of gesdd's bidiagonalisation)
elimination loop (:146-175)              ``basq_car_eliminate_f64``
re-weight + compaction (:107-130)        ``basq_reweight_compact_f64`` (closed-form destinations)
=======================================  ==========================================================

Block-sum strategies (what differs between the kernel kinds is ONLY how a round's message is formed):

* ``FusedSums``  -- structured kernels (``basq_amd.kernels``): the fused pairwise kernel, per residue class of the block
  index where the round structure allows it (plain BASQ rounds of RBF / Matern / posterior / WSABI-L kernels), with the
  SOBER remainder columns and WSABI-M's squared-covariance term as explicit add-ons;
* ``OpaqueSums`` -- any callable: dense chunks (or the reference's own block-by-block calls) through
  ``basq_dense_blocksum_f64``.

Which combination a batch runs is decided ONCE, in ``Plan.of`` -- unsupported combinations raise there.

The batch never blocks: wherever the host has to wait for the GPU, ``steps()`` yields the event, and the engine decides
whether to block on it (one batch) or to advance another batch meanwhile (``RecombinationEngine.run_many``).
"""
from __future__ import annotations

import dataclasses
import time
import warnings
from dataclasses import dataclass

import torch

from . import _config as cfg
from . import _epochs
from ._basis import (_lapack_threads, _mm_splitk, _ShardedProducts, _skip_test_matrix_draw, _Timer, make_cov_psd,
                     nystrom_basis_steps)
from ._lib import ROLE_A, ROLE_B
from ._partition import (RoundGeometry, choose_chunks, initial_shards, local_blocks, next_shard, survivors_before)


class ReductionTimeout(RuntimeError):
    """A cluster reduction kernel gave up waiting for its sibling work-groups (status 2)."""


class _NoWait:
    """An event that has already fired (ops without events: the CPU stand-in of the tests)."""

    def synchronize(self):
        pass

    def query(self):
        return True


def _recorded_event(ops):
    ev = ops.record_event(False) if getattr(ops, "name", "") == "hip" else None
    return ev if ev is not None else _NoWait()


def classes_for(nb_global: int) -> int:
    """Number of residue classes (a power of two, 1 = none) for an evaluation over ``nb_global`` full blocks."""
    if not cfg.CLASS_SUMS:
        return 1
    c = cfg.MAX_CLASSES
    while c > 1 and nb_global < 4 * c:           # at least four blocks per class (the classes are also the chunks)
        c //= 2
    return c


def late_split(off: int, Rl: int, n_full: int, S: int, n_chunks: int, n_late: int):
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


@dataclass(frozen=True)
class Plan:
    """Which code paths a batch takes -- every flag combination that exists is spelled out here."""
    opaque: bool          # kernel is a bare callable (dense chunks)
    sober: bool           # SOBER/_rchq.py semantics (init weights, remainder double count, early stop)
    warp: str             # "none" | "wsabil" | "wsabim"
    posterior: bool       # GP posterior correction folded into the contraction matrix
    objective: bool       # SOBER objective row (single reduction only)
    classes: bool         # residue-class block sums (epochs of regrouped rounds)
    async_rounds: bool    # rounds driven by the device-resident descriptor (no host wait per round)
    world: int

    @staticmethod
    def of(kernel, variant, objective, comm, ops, trace, n_sets=0, owner=None):
        """``n_sets`` = 2 * (min(num_pts - 1, m) + 1), the width of every round's reduction (0: unknown, no limit)."""
        if variant not in ("basq", "sober"):
            raise ValueError(variant)
        sober = variant == "sober"
        opaque = bool(getattr(kernel, "opaque", False))
        warp = "none" if opaque else kernel.warp
        posterior = (not opaque) and kernel.posterior is not None
        if opaque and sober:
            raise NotImplementedError("the SOBER variant needs a structured kernel (basq_amd.kernels)")
        if getattr(kernel, "jitter", 0.0) != 0.0:
            # wsabil/wsabim_kernel add `jitter` to entries [k][k] of every block (_wsabi.py:223,247), UNweighted by the
            # warped means; the reference hard-codes jitter = 0 (_wsabi.py:56) and the fused path carries no such term
            raise NotImplementedError("WsabiKernel.jitter != 0 is not supported by the fused recombination path")
        if objective is not None:
            if not sober:
                raise ValueError("an objective is part of the SOBER variant only")
            if comm.world > 1 or posterior or warp != "none":
                raise NotImplementedError("objective row: single process, stationary kernels only")
        # class sums need rounds that are plain block sums of a structured kernel and keep exactly half of the sets
        # (SOBER's first count of the ragged remainder, SOBER/_rchq.py:127-135, rides along as one more irregular chunk)
        # and WSABI-M's squared covariance as one more per-pair block sum -- its likelihood-noise cross terms, which sit on
        # one Nystrom row per candidate, are evaluated per round: FusedSums.wsabim_class_round)
        classes = cfg.CLASS_SUMS and not opaque
        # descriptor-driven rounds: the same restriction + the ops must provide the *_geo entries; a trace that
        # synchronises per phase (host timers, per-round tensors) needs the round-by-round loop
        traced_sync = trace is not None and (trace.host_sync or trace.keep_tensors)
        # ... and they call the GPU null-space / elimination kernels directly: wider reductions than those hold
        # (2 * num_pts > 1024) and the host-LAPACK route (GPU_NULLSPACE off) take the round-by-round loop
        gpu_reduction = cfg.GPU_NULLSPACE and n_sets <= getattr(ops, "NULLSPACE_MAX_M", 1 << 30)
        async_rounds = (cfg.ASYNC_ROUNDS and not opaque and objective is None
                        and (warp != "wsabim" or hasattr(ops, "blocksum_sq_geo"))
                        and hasattr(ops, "round_next") and not traced_sync and gpu_reduction
                        and (comm.world == 1 or cfg.REPLICATED_REDUCTION or owner is not None))
        return Plan(opaque, sober, warp, posterior, objective is not None, classes, async_rounds, comm.world)


class Batch:
    """State of one recombination in flight; ``steps()`` is the generator the engine drives."""

    def __init__(self, ops, comm, pts_local, gid0: int, n_total: int, pts_nys, num_pts: int, kernel, trace=None,
                 variant: str = "basq", init_weights=None, objective=None, pipelined: bool = False, owner: int | None = None,
                 draw_on_owner: bool = False):
        """``pts_local`` = this rank's contiguous slice ``[gid0, gid0 + len)`` of the pool.

        ``variant="basq"`` follows ``BASQ/_rchq.py`` (uniform start weights, ``init_weights`` ignored);
        ``variant="sober"`` follows ``SOBER/_rchq.py``: ``init_weights`` (this rank's slice of them)
        are honoured and zero-weight points dropped, the Nystrom Gram goes through ``make_cov_psd``, the ragged
        remainder is additionally added to sets ``0..N_rest-1`` (:127-135), and an elimination that finds no
        positive entry stops early instead of failing (:240-242).

        ``objective`` (sober only): ``-calc_obj(pts_rec)`` (``SOBER/_rchq.py:67-69``), one value per local candidate.
        The reference can only execute its objective branch when the pool fits a single reduction (``:77-104``); for
        larger pools it raises at ``:140-142`` -- and so does this engine.

        ``pipelined``: other batches are in flight on other streams (nothing is deferred behind the range finder then:
        their kernels fill that gap).

        ``owner`` (several ranks): the rank that runs this batch's per-round reductions (null space + elimination) and
        broadcasts their outcome -- ``3 M + 1`` doubles, stream-ordered -- instead of every rank repeating them.  With k >= G
        batches in flight, owners dealt round-robin, every GPU carries 1/G of the chains (``RecombinationEngine.run_many``);
        None = every rank reduces the gathered message itself (no broadcast: the choice for ONE batch at a time).
        """
        if n_total >= 2 ** 31:
            raise ValueError("pool sizes >= 2^31 are not supported")
        self.ops, self.comm, self.trace = ops, comm, trace
        self.notes = []                                  # conditions the caller should know about (-> warnings)
        d = int(pts_nys.shape[1]) if pts_nys.dim() == 2 else 0
        if not getattr(kernel, "opaque", False) and not kernel.base.fits_packed_rows(d):
            # envelope: the fused kernels pack a point into <= 40 doubles (d <= 38).  The reference accepts any d
            # (_rchq.py:4-25), so wider points degrade to the dense path -- the structured kernel, which is a callable with
            # the reference's semantics, evaluated block by block like any other callable -- instead of raising
            from .kernels import CallableKernel

            self.notes.append(f"dimension {d} exceeds the fused kernels' packed-row limit ({kernel.base.__class__.__name__}: "
                              "d <= 38); the kernel is evaluated densely on the device (basq_dense_blocksum_f64 path)")
            kernel = CallableKernel(kernel)
        self.kernel, self.variant = kernel, variant
        self.pts_local_in, self.pts_nys_in = pts_local, pts_nys
        self.gid0, self.n_total, self.num_pts = int(gid0), int(n_total), int(num_pts)
        self.init_weights, self.objective = init_weights, objective
        self.pipelined = pipelined
        self.owner = None if (owner is None or comm.world == 1) else int(owner) % comm.world
        # the Gaussian draw of the sharded range finder on the owner alone (run_many: every job carries a seed)
        self.draw_on_owner = bool(draw_on_owner) and self.owner is not None
        m_nys = int(pts_nys.shape[0]) if pts_nys.dim() >= 1 else 0
        self.plan = Plan.of(kernel, variant, objective, comm, ops, trace, n_sets=2 * (min(int(num_pts) - 1, m_nys) + 1),
                            owner=self.owner)

    # ------------------------------------------------------------------------------------------------
    # the step generator
    # ------------------------------------------------------------------------------------------------
    def steps(self):
        """-> ``(idx int64[<=num_pts] ascending, w float64)`` on the ops device (identical on every rank)."""
        try:
            return (yield from self._steps())
        finally:
            self._release()

    def _release(self):
        """Drop every reference the batch holds (the strategy object points back at the batch: without this the cycle
        keeps hundreds of MB of device buffers alive until Python's cycle collector runs, and the next batch allocates
        afresh instead of reusing them)."""
        sums = self.__dict__.get("sums")
        if sums is not None:
            sums.b = None
        keep = ("notes", "plan", "drew_test_matrix")
        for k in list(self.__dict__):
            if k not in keep:
                self.__dict__[k] = None

    def _steps(self):
        ops = self.ops
        if self.n_total == 0:                                   # empty pool: nothing to select (the reference returns [])
            dev = getattr(ops, "device", "cpu")
            return torch.empty(0, dtype=torch.int64, device=dev), torch.empty(0, dtype=torch.float64, device=dev)
        t_all = time.perf_counter()
        self._prepare_operands()
        self._init_candidates()
        pre, late = self._prelaunch_round1()
        t0 = time.perf_counter()
        U = yield from self._basis_steps(late)
        self._trace_phase("basis", t0)
        if self.trace is not None and self.trace.keep_tensors:
            self.trace.U = U.clone()
        self._extend_basis(U)
        self._trace_phase("setup", t0)
        if self.plan.async_rounds and self.R > self.S:
            # False | True (a round violated the plan) | "basis"
            if _epochs.eligible(self):
                outcome = yield from _epochs.async_rounds_columns(self, pre)     # no pairwise evaluation inside an epoch
            else:
                outcome = yield from self._async_rounds(pre)
            pre = None
            if outcome == "basis":
                # the range finder's pivot flag arrived with the descriptor table: the rounds ran on a basis that is not
                # orthonormal to round-off -- recompute it on the host, from the same Gaussian draw, and start over
                self._recover_basis()
                self._init_candidates()
            elif outcome:
                # an elimination failed or did not keep half of the sets while regrouped class messages were already
                # enqueued: repeat the ROUNDS one read-back at a time -- same basis (no second draw from the generator),
                # same results as that loop alone
                self.notes.append("descriptor-driven rounds hit a round that did not keep exactly half of the sets; the "
                                  "rounds were repeated one read-back at a time")
                self._init_candidates()
        if self._basis_bad is not None:
            # (a batch without descriptor-driven rounds: the flag is read on its own, before the first round uses the basis)
            flag, ready = ops.to_host_async(self._basis_bad, "basis_flag")
            yield ready
            if float(flag[0]) != 0.0:
                self._recover_basis()                            # (the round-1 block sums in `pre` do not depend on the basis)
            self._basis_bad = None
        idx, w = yield from self._sync_rounds(pre)
        self._trace_phase("total", t_all)
        for msg in self.notes:
            warnings.warn("basq_amd.recombination: " + msg, RuntimeWarning, stacklevel=3)
        return idx, w

    def _trace_phase(self, key, t0):
        if self.trace is not None:
            if self.trace.host_sync:
                self.ops.synchronize()
            self.trace.add_time(key, time.perf_counter() - t0)

    # ------------------------------------------------------------------------------------------------
    # set-up
    # ------------------------------------------------------------------------------------------------
    def _prepare_operands(self):
        """Nystrom-side operands of the block sums (no dependence on the basis)."""
        ops, kernel, plan = self.ops, self.kernel, self.plan
        self.pts_nys = pts_nys = ops.to_device(self.pts_nys_in, torch.float64)
        self.pts_local = ops.to_device(self.pts_local_in, torch.float64)
        self.m, self.d = m, d = pts_nys.shape
        self.base = None if plan.opaque else kernel.base
        self.post = kernel.posterior if plan.posterior else None
        self.spec = None if plan.opaque else self.base.spec(d)
        if plan.posterior:
            # the message of a posterior batch is k - k(., X) W k(X, .): a cancellation that amplifies kernel-value errors
            # by up to the conditioning of the observation Gram -- its block sums take the 1e-17 exponential
            self.spec = dataclasses.replace(self.spec, accurate_exp=True)
        self.kp = d if plan.opaque else ops.kp(d)
        self.kscale = 1.0 if plan.opaque else self.spec.outputscale
        self.center = None if plan.opaque else ops.col_mean(pts_nys)
        self.q = q = min(self.num_pts - 1, m)                   # rank of svd_lowrank's output (reduced QR clips at m)
        self.s = q + 1
        self.S = 2 * self.s                                     # :50
        nys_rows = [pts_nys]
        self.diag_noise, self.n_obs = 0.0, 0
        if self.post is not None:
            cond = self.post.condition_number() if hasattr(self.post, "condition_number") else 0.0
            if cond > self.post.COND_WARN:
                self.notes.append(f"the observation Gram of the GP posterior is ill-conditioned (cond ~ {cond:.1e}): the "
                                  "posterior covariance is a catastrophic cancellation, and the reference's own selection is "
                                  "not reproducible to the last ulp in this regime (DESIGN.md section 2)")
            self.Xo = ops.to_device(self.post.Xobs, torch.float64)
            self.n_obs = self.Xo.shape[0]
            nys_rows.append(self.Xo)
            self.diag_noise = self.post.noise
        self.m_ext = m + self.n_obs
        self.q_ext = q
        self.wrow = 0
        self.zero_row_idx = None
        if plan.warp != "none" and self.diag_noise != 0.0:
            # an all-zero packed row has kernel value 1 with every candidate: its block sum is the
            # kernel-weighted set weight needed by the diagonal-noise term of wsabil_kernel
            self.zero_row_idx = self.m_ext
            self.m_ext += 1
            self.q_ext = q + 1
            self.wrow = q + 1
        self.nys_ext = None
        if not plan.opaque:
            nys_cat = torch.cat(nys_rows, 0) if len(nys_rows) > 1 else pts_nys
            nys_ext = ops.pack(self.spec, nys_cat, self.center, ROLE_A, pad_rows_to=64)
            if nys_ext.shape[0] < ((self.m_ext + 63) // 64) * 64:
                nys_ext = torch.cat([nys_ext, ops.zeros(64, self.kp)], 0)
            if self.wrow:
                nys_ext[self.zero_row_idx].zero_()
            self.nys_ext = nys_ext
        self.exact_blocks = plan.opaque and kernel.resolve_mode(ops, pts_nys, self.S)
        self.sums = OpaqueSums(self) if plan.opaque else FusedSums(self)

    def _init_candidates(self):
        """Candidate state of round 1: ``cand`` (packed rows, or raw rows for a callable), ``mu``, ``gid``, ``wx`` of this
        rank's shard ``[off, off + Rl)`` of the R live positions.  Re-runnable (the rounds can be repeated)."""
        ops, comm, plan = self.ops, self.comm, self.plan
        pts_local = self.pts_local
        Rl = pts_local.shape[0]
        self.cand = (pts_local if Rl > 0 else ops.zeros(1, self.d)) if plan.opaque \
            else ops.pack(self.spec, pts_local, self.center, ROLE_B)
        self.mu, self.gid = ops.init_state(Rl, self.gid0, self.n_total)
        self.wx = None
        if plan.warp != "none":
            self.wx = self.kernel.mean(ops, pts_local, self.center) if Rl > 0 else ops.empty(1)
        self.off, self.R, self.Rl = self.gid0, self.n_total, Rl
        self.obj_full = self.obj_live = None
        if plan.objective:
            self.obj_full = self.obj_live = ops.to_device(self.objective, torch.float64).reshape(-1)
            if self.obj_full.shape[0] != Rl:
                raise ValueError("objective must have one entry per candidate")
        if plan.sober and self.init_weights is not None:
            # SOBER/_rchq.py:60-64: start from the given weights, drop the zero-weight points up front
            w0 = ops.to_device(self.init_weights, torch.float64)
            if w0.shape[0] != Rl:
                raise ValueError("init_weights must have one entry per local candidate")
            nz = torch.nonzero(w0 != 0).reshape(-1)
            self.cand, self.mu, self.gid = self.cand[nz].contiguous(), w0[nz].contiguous(), self.gid[:Rl][nz].contiguous()
            if self.obj_live is not None:
                self.obj_live = self.obj_live[nz].contiguous()
            if self.wx is not None:
                self.wx = self.wx[nz].contiguous()
            Rl = int(nz.numel())
            counts = torch.tensor([float(Rl)], dtype=torch.float64, device=self.mu.device)
            if comm.world > 1:
                counts = comm.all_gather(counts).reshape(-1)
            counts = [int(v) for v in counts.cpu()]
            self.off, self.R, self.Rl = sum(counts[:comm.rank]), sum(counts), Rl
            if Rl == 0:                                         # keep pointers valid for empty shards
                self.cand, self.mu, self.gid = ops.zeros(1, self.kp), ops.zeros(1), ops.zeros(1, dtype=torch.int64)
        self.cls = None                                         # inherited class MESSAGES: dict(M [C + 1, rows, S], C, reg_blocks)
        self.R_lo = self.R                                      # lower bound of R: the class plan follows it on every path

    def _prelaunch_round1(self):
        """Round-1 block sums are queued BEFORE the basis: they do not depend on U, and the host's RNG draw for the range
        finder then overlaps with the largest kernel of the batch.  -> ``(pre, late)``: the evaluation record and the
        deferred launches (run behind the range finder's GPU work), either may be None."""
        if self.R > self.S and not self.plan.opaque:
            pre = self.sums.evaluate(RoundGeometry.of(self.R, self.S), self.S, defer_last=True)
            return pre, pre[5]
        return None, None

    def _basis_steps(self, late):
        """Nystrom Gram + range finder -> ``U [q, m]`` (identical on every rank).  The range finder does not wait for the host
        (``_basis.nystrom_basis_steps``): its pivot flag -- ``self._basis_bad``, a device scalar, identical on every rank -- is
        read at the batch's first read-back, and ``_recover_basis`` recomputes a flagged basis on the host."""
        ops, comm, kernel, trace = self.ops, self.comm, self.kernel, self.trace
        m, q, pts_nys = self.m, self.q, self.pts_nys
        self._basis_bad = self._basis_fallback = None
        self._basis_on_rank0 = False
        if cfg.SHARDED_BASIS and comm.world > 1 and not self.plan.sober and not self.plan.opaque:
            # every rank builds its row block of the Gram matrix and takes part in the range finder (no broadcast of U)
            shards = initial_shards(m, comm.world)
            r0, mr = shards[comm.rank]
            with _Timer(ops, trace, "basis.gram"):
                A_rows = kernel.dense(ops, pts_nys[r0:r0 + mr].contiguous(), pts_nys, self.center, diag_offset=r0) \
                    if mr else ops.zeros(0, m)
            prod = _ShardedProducts(ops, comm, A_rows, shards, m, drawer=self.owner if self.draw_on_owner else 0,
                                    lockstep=not self.draw_on_owner)
            self.drew_test_matrix = comm.rank == prod.drawer or prod.lockstep
            res = yield from nystrom_basis_steps(ops, prod, self.num_pts - 1, trace, overlap=late, notes=self.notes)
            if res.bad is not None:
                self._basis_bad, self._basis_fallback = res.bad.to(torch.float64), res.fallback
            return res.U
        self._basis_on_rank0 = True
        bad = None
        if comm.rank == 0:
            with _Timer(ops, trace, "basis.gram"):
                A = kernel.dense(ops, pts_nys, pts_nys, self.center)
                if self.plan.sober:
                    A = make_cov_psd(A)
            res = yield from nystrom_basis_steps(ops, A, self.num_pts - 1, trace, overlap=late, notes=self.notes)
            del A
            U, bad, self._basis_fallback = res.U, res.bad, res.fallback
            assert U.shape[0] == q
        else:
            U = ops.empty(q, m)
            _skip_test_matrix_draw(ops, m, self.num_pts - 1)    # keep this rank's global generator in step with rank 0
            if late is not None:
                late()                                          # runs while rank 0 finishes the basis
            if cfg.BASIS_SVD and cfg.GPU_RANGE_FINDER and self.num_pts - 1 <= m:
                # (round 3's path only) rank 0 yields exactly once for the range finder's q x q SVD: yield at the same point,
                # so that ``run_many`` resumes the batches -- and every rank enqueues its collectives -- in ONE order
                yield _recorded_event(ops)
        if comm.world > 1:
            # U and the pivot flag in ONE broadcast: row q of the buffer carries the flag
            buf = ops.empty(q + 1, m)
            if comm.rank == 0:
                buf[:q] = U
                buf[q].zero_()
                if bad is not None:
                    buf[q, 0:1] = bad.to(torch.float64)
            comm.broadcast(buf)
            U = buf[:q]
            self._basis_bad = buf[q, 0:1] if (cfg.GPU_RANGE_FINDER and not cfg.BASIS_SVD and self.num_pts - 1 <= m) else None
        elif bad is not None:
            self._basis_bad = bad.to(torch.float64)
        return U

    def _recover_basis(self):
        """The range finder's pivot flag was set (a numerically rank-deficient panel): the basis again, by host Householder QR
        and the reference's SVD from the same Gaussian draw -- every rank takes part (the sharded products gather the Gram
        matrix; a basis computed on rank 0 is broadcast again)."""
        ops, comm = self.ops, self.comm
        if not self._basis_on_rank0:
            U = self._basis_fallback()
        else:
            U = self._basis_fallback() if comm.rank == 0 else ops.empty(self.q, self.m)
            if comm.world > 1:
                U = U.contiguous()
                comm.broadcast(U)
        if self.trace is not None and self.trace.keep_tensors:
            self.trace.U = U.clone()
        self._basis_bad = None
        self._extend_basis(U)

    def _extend_basis(self, U):
        """Extended contraction matrix: posterior correction / warping folded in by linearity --
        ``U @ sum mu k_post(pt, x) = [U, -U K(pt,X) W] @ sum mu k([pt; Xobs], x)``; WSABI-L's ``mu(x)`` factors are a
        per-candidate weight (``wx``) and a column scaling of ``U``."""
        ops, plan, q, m = self.ops, self.plan, self.q, self.m
        self.U = U
        Um = U
        if plan.warp != "none":
            mu_pt = self.kernel.mean(ops, self.pts_nys, self.center)
            Um = (U * mu_pt.unsqueeze(0)).contiguous()
        U_cols = [Um]
        self.bmatT = None
        if self.post is not None:
            W = ops.to_device(self.post.W, torch.float64)
            # [m, n_obs] and [q, n_obs], once per batch, on the own tall-skinny kernel (n_obs <= 208; wider: library GEMM)
            Bmat = _mm_splitk(ops, self.base.dense(ops, self.pts_nys, self.Xo, self.center), W, 1)
            U_cols.append(-_mm_splitk(ops, Um, Bmat, 8))
            if plan.warp == "wsabim":
                # B^T, zero-padded to whole MFMA fragments: the A operand of the fused squared-covariance block sums
                self.bmatT = ops.zeros(((self.n_obs + 3) // 4) * 4, ((m + 63) // 64) * 64)
                self.bmatT[:self.n_obs, :m] = Bmat.t()
        U_ext = torch.cat(U_cols, 1) if len(U_cols) > 1 else Um
        if self.wrow:
            sel = ops.zeros(1, self.m_ext)
            U_ext = torch.cat([torch.cat([U_ext, ops.zeros(q, 1)], 1), sel], 0)
            U_ext[q, self.zero_row_idx] = 1.0 / self.kscale
        self.U_ext = U_ext.contiguous()
        self.diagU = Um if self.diag_noise != 0.0 else None

    # ------------------------------------------------------------------------------------------------
    # rounds without a host round trip
    # ------------------------------------------------------------------------------------------------
    def _async_rounds(self, pre):
        """The rounds that are CERTAINLY not the final one, enqueued without waiting for the GPU.

        The survivor count of a round depends on the data through two facts only (how many sets were kept, whether the
        last set -- owner of the ragged tail -- is one of them), so the next round's geometry, INCLUDING this rank's shard
        of it, is a closed form a one-thread kernel evaluates into a device-resident descriptor; every launch of the round
        reads its candidate range from there, and the per-round exchange of a multi-rank run (all-gather of the
        ``(q+1) x S`` messages) is stream-ordered like everything else.  The host enqueues all rounds whose lower bound of
        the survivor count exceeds S, then reads the descriptor once.  -> True when the descriptor carries the violation
        flag (the caller repeats the rounds one read-back at a time)."""
        ops, comm, trace = self.ops, self.comm, self.trace
        S, s, q, m, m_ext, q_ext = self.S, self.s, self.q, self.m, self.m_ext, self.q_ext
        spec, nys_ext, U_ext, kscale, kp = self.spec, self.nys_ext, self.U_ext, self.kscale, self.kp
        diag_noise, diagU, wrow = self.diag_noise, self.diagU, self.wrow
        multi = comm.world > 1
        owner = self.owner                                       # None: every rank reduces; else: that rank + a broadcast
        n_keep_exp = s                                           # a regular round keeps s = S/2 sets
        reg_hi0 = (pre[4] * S) if (pre is not None and pre[3] >= 2) else 0
        geo_t = ops.geo_init(64, self.R, S, reg_hi0, self.off, self.Rl)
        r = 0
        R_lo = R_up = self.R
        Rl_up = self.Rl                                          # upper bound of this rank's shard (sizes launches / buffers)
        plan_C = None
        cls = None
        records = []                                             # per enqueued round, for the trace: (info|kept buffer)
        cand, mu, gid, wx = self.cand, self.mu, self.gid, self.wx
        n_extra = self.sums.n_extra
        # WSABI-M (_wsabi.py:240-242): the squared covariance is one more per-pair block sum, added to the class messages; its
        # likelihood-noise cross terms -- one Nystrom row per candidate, a different one every round -- are one more message PART
        wsm = self.plan.warp == "wsabim"
        noise_slot = 1 if (wsm and diag_noise != 0.0) else 0
        rows_msg = q_ext + 1

        def wsabim_kobs():
            """``outputscale * k(Xobs, x_p)`` of this rank's live candidates (sized by the upper bound of their number)."""
            n4, width = self.bmatT.shape[0], max(Rl_up, 1)
            kobs = ops.empty(n4, width)
            if n4 != self.n_obs:
                kobs[self.n_obs:].zero_()
            ops.gram_into(spec, nys_ext[m:m + self.n_obs], self.n_obs, cand, width, kobs)
            return kobs

        def wsabim_classes(Mc_, C_, fresh):
            """The squared term of a class round (``FusedSums.wsabim_class_round`` with the ranges read from the descriptor)."""
            kobs = wsabim_kobs()
            n_sq = (C_ if fresh else 0) + n_extra
            Epart = ops.empty(n_sq, m, S)
            k = 0
            if fresh:
                ops.blocksum_sq_geo(spec, nys_ext, m, cand, mu, g_row, 1, S, C_, self.bmatT, kobs, self.n_obs, 0.0,
                                    class_mod=C_, class0=0, out=Epart[:C_])
                k = C_
            ops.blocksum_sq_geo(spec, nys_ext, m, cand, mu, g_row, 2, S, 1, self.bmatT, kobs, self.n_obs, 0.0, out=Epart[k:k + 1])
            if n_extra == 2:
                ops.blocksum_sq_geo(spec, nys_ext, m, cand, mu, g_row, 4, S, 1, self.bmatT, kobs, self.n_obs, 0.0,
                                    out=Epart[k + 1:k + 2])
            Me = ops.project_chunks(self.U, q, m, Epart, ops.zeros(n_sq, S), n_sq, S, 1.0)
            slots = Mc_[:C_ + n_extra] if fresh else Mc_[C_:C_ + n_extra]
            slots[:, 1:q + 1] += Me[:, 1:q + 1]
            if noise_slot:
                val = ops.cov_diag_geo(spec, nys_ext, m, cand, g_row, Rl_up, S, self.bmatT, kobs, self.n_obs, diag_noise)
                ops.sq_noise_part_geo(mu, val, g_row, self.U, q, m, S, rows_msg, n_extra == 2, Mc_[C_ + n_extra])

        def wsabim_plain(msg_, n_ch):
            """... and of a round without classes: the kernel carries the noise itself (``FusedSums.wsabim_square_term``)."""
            kobs = wsabim_kobs()
            Epart = ops.blocksum_sq_geo(spec, nys_ext, m, cand, mu, g_row, 3, S, n_ch, self.bmatT, kobs, self.n_obs, diag_noise)
            E = Epart[0] if n_ch == 1 else ops.sum_parts(Epart)
            if n_extra == 2:                                     # SOBER's first count of the remainder: the whole kernel again
                E = E + ops.blocksum_sq_geo(spec, nys_ext, m, cand, mu, g_row, 4, S, 1, self.bmatT, kobs, self.n_obs, diag_noise)[0]
            msg_[1:q + 1] += _mm_splitk(ops, self.U, E, 8)

        def tail_block_geo(Xslot, totslot):
            """SOBER's first count of the remainder (descriptor geometry: ``geo_mode`` 4); no set weight is added there."""
            ops.blocksum_geo(spec, nys_ext, m_ext, cand, mu, wx, g_row, 4, S, 1, out=(Xslot, totslot))
            totslot.zero_()

        while R_lo > S:
            g_row = geo_t[r]
            Mc, C_cur, parts = None, 1, None
            if cls is not None:                                  # inside an epoch: regrouped class messages + the rest
                Mc, C_cur = cls["M"], cls["C"]
                Xirr, totirr = ops.empty(n_extra, m_ext, S), ops.empty(n_extra, S)
                self.sums.timed_geo(r, 2, 1.0, lambda: ops.blocksum_geo(spec, nys_ext, m_ext, cand, mu, wx, g_row, 2, S, 1,
                                                                        out=(Xirr[:1], totirr[:1])))
                if n_extra == 2:
                    tail_block_geo(Xirr[1:2], totirr[1:2])
                ops.project_chunks(U_ext, q_ext, m_ext, Xirr, totirr, n_extra, S, kscale, out=Mc[C_cur:C_cur + n_extra])
                if wsm:
                    wsabim_classes(Mc, C_cur, fresh=False)
                parts = Mc
            else:
                if pre is not None:                              # round 1: launched before the basis, host geometry
                    Xpart, totpart, n_chunks, C_cur = pre[:4]
                    pre = None
                else:
                    C_cur = plan_C if plan_C is not None else 1
                    if C_cur >= 2:
                        n_chunks = C_cur + n_extra
                        Xpart, totpart = ops.empty(n_chunks, m_ext, S), ops.empty(n_chunks, S)
                        self.sums.timed_geo(r, 1, 1.0, lambda: ops.blocksum_geo(
                            spec, nys_ext, m_ext, cand, mu, wx, g_row, 1, S, C_cur, out=(Xpart[:C_cur], totpart[:C_cur]),
                            class_mod=C_cur))
                        self.sums.timed_geo(r, 2, 1.0, lambda: ops.blocksum_geo(
                            spec, nys_ext, m_ext, cand, mu, wx, g_row, 2, S, 1, out=(Xpart[C_cur:C_cur + 1],
                                                                                     totpart[C_cur:C_cur + 1])))
                        if n_extra == 2:
                            tail_block_geo(Xpart[C_cur + 1:], totpart[C_cur + 1:])
                    else:
                        n_plain = choose_chunks(max(R_lo // S // comm.world, 1), m_ext, S, kp // 4)
                        n_chunks = n_plain + (n_extra - 1)
                        Xpart, totpart = ops.empty(n_chunks, m_ext, S), ops.empty(n_chunks, S)
                        self.sums.timed_geo(r, 3, 1.0, lambda: ops.blocksum_geo(spec, nys_ext, m_ext, cand, mu, wx, g_row, 3, S,
                                                                                n_plain, out=(Xpart[:n_plain], totpart[:n_plain])))
                        if n_extra == 2:
                            tail_block_geo(Xpart[n_plain:], totpart[n_plain:])
                if C_cur >= 2:
                    Mc = ops.empty(n_chunks + noise_slot, rows_msg, S)
                    ops.project_chunks(U_ext, q_ext, m_ext, Xpart, totpart, n_chunks, S, kscale, out=Mc[:n_chunks])
                    if wsm:
                        wsabim_classes(Mc, C_cur, fresh=True)
                    parts = Mc
                else:
                    parts = ops.project(U_ext, q_ext, m_ext, Xpart, totpart, n_chunks, S, kscale).unsqueeze(0)
                    if wsm:
                        wsabim_plain(parts[0], max(1, n_chunks - (n_extra - 1)))
                del Xpart, totpart
            rows = parts.shape[1]
            if diag_noise != 0.0:
                # predictive_covariance's noise on the ragged tail block (entries [k][k], tail point k x Nystrom row
                # k): one more message row carries the tail weights; its length is known on the device only, so the
                # row is always there (all zeros without a tail: the extra terms vanish)
                buf = ops.empty(1, rows + 1, S)
                ops.sum_parts(parts, out=buf[0, :rows])
                ops.tail_weights_geo(mu, wx, g_row, S, buf[0, rows])
                if multi:
                    buf = comm.all_gather(buf[0])
                fin = (buf, buf.shape[0], rows + 1, q, S, diagU, m, min(m, S), diag_noise, wrow, rows, min(m, S), g_row)
            else:
                if multi:
                    parts = comm.all_gather(ops.sum_parts(parts) if parts.shape[0] > 1 else parts[0])
                fin = (parts, parts.shape[0], rows, q, S, None, m, min(m, S), 0.0, 0, 0, 0, None)
            res, rv = ops.reduction_result(S) if owner is not None else (None, None)
            if owner is None or comm.rank == owner:
                ev_c = ops.record_event() if self.sums._timing() else None
                XcarT, tot = ops.finalize(*fin, tot_out=None if rv is None else rv["tot"])
                PhiT = ops.nullspace(XcarT, s, S)
                keep_rank, kept, w_star, info = ops.car_eliminate(PhiT, tot, S, s, out=rv)
                if ev_c is not None:                             # the round's chain of single-work-group kernels
                    trace.chain_events.append((ev_c, ops.record_event()))
            if owner is not None:
                # the outcome of the owner's reduction (w_star | tot | info, kept, keep_rank: 3 S + 1 doubles), stream-ordered
                comm.broadcast(res, src=owner)
                keep_rank, kept, w_star, info, tot = rv["keep_rank"], rv["kept"], rv["w_star"], rv["info"], rv["tot"]
            if trace is not None:
                records.append(ops.info_kept_buffer(info, kept))
            # bounds of the next survivor count; the class plan of the next fresh evaluation follows the lower one
            R_lo_n = (R_lo // S) * n_keep_exp
            R_up_n = (R_up // S) * n_keep_exp + (S - 1)
            # this rank's shard [off, off + Rl): at most ceil(Rl / S) + 1 blocks touch it, each keeps n_keep_exp; + the tail
            Rl_up_n = min(R_up_n, (-(-Rl_up // S) + 1) * n_keep_exp + (S - 1)) if multi else R_up_n
            cls = None
            plan_C = None
            if Mc is not None and C_cur >= 2:
                # next round's class messages AND its descriptor, one launch (both read the elimination's outcome)
                Mn = ops.empty(C_cur // 2 + n_extra + noise_slot, Mc.shape[1], S)
                ops.regroup_round_next(Mc[:C_cur], kept, w_star, tot, Mn[:C_cur // 2], g_row, info, keep_rank, S, -1, True,
                                       geo_t[r + 1])
                cls = dict(M=Mn, C=C_cur // 2, reg_blocks=None)
            else:
                plan_C = classes_for(R_lo_n // S) if self.plan.classes else 1
                ops.round_next(g_row, info, keep_rank, S, plan_C if plan_C >= 2 else 0, True, geo_t[r + 1])
            cand, mu, gid, wx = ops.reweight_compact_geo(cand, mu, gid, wx, g_row, geo_t[r + 1], info, Rl_up, S, kp,
                                                         keep_rank, w_star, tot, Rl_up_n, n_keep_exp)
            r += 1
            R_lo, R_up, Rl_up = R_lo_n, R_up_n, Rl_up_n
        if multi and owner is None:
            # every rank ran its own reductions: a cluster-kernel time-out (status 2) is local to ONE rank, and the ranks
            # must agree on repeating the rounds -- the flag becomes the maximum over the ranks
            flags = comm.all_gather(geo_t[r, 3:4].to(torch.float64))
            geo_t[r, 3:4] = flags.max().to(torch.int64).reshape(1)
        bad64 = (self._basis_bad != 0).to(torch.int64) if self._basis_bad is not None else geo_t[0, 3:4] * 0
        flat, ready = ops.to_host_async(torch.cat([geo_t[:r + 1].reshape(-1), bad64.reshape(1)]), "geo_table")
        yield ready                                              # the ONE wait of the asynchronous rounds
        table = flat[:-1].view(r + 1, 8)
        if self._basis_bad is not None:
            self._basis_bad = None
            if int(flat[-1]) != 0:
                return "basis"
        row = table[r].tolist()
        if row[3] != 0:
            return True
        if trace is not None:
            self._trace_async_rounds(table, records, r)
        self.cand, self.mu, self.gid, self.wx = cand, mu, gid, wx
        self.R, self.off, self.Rl = int(row[0]), int(row[6]), int(row[7])
        self.R_lo = R_lo
        if cls is not None:
            cls["reg_blocks"] = int(row[2]) // S
            if noise_slot:                                       # (the round-by-round loop adds that part by itself)
                cls["M"] = cls["M"][:cls["C"] + n_extra]
        self.cls = cls
        return False

    def _trace_async_rounds(self, table, records, r):
        """Round records of the descriptor-driven rounds, read back after the fact (one copy per enqueued round)."""
        ops, trace, S = self.ops, self.trace, self.S
        for k in range(r):
            g = table[k].tolist()
            ik = ops.to_host(records[k], "head").tolist()
            trace.rounds.append(dict(R=int(g[0]), S=S, nb=int(g[4]), n_tail=int(g[5]), kept=ik[2:2 + ik[0]]))
        self.sums.resolve_geo_events(table)

    # ------------------------------------------------------------------------------------------------
    # rounds with one read-back each (the last two or three of a batch; every round of a traced / SOBER / WSABI-M /
    # opaque batch)
    # ------------------------------------------------------------------------------------------------
    def _sync_rounds(self, pre):
        ops, comm, trace, plan = self.ops, self.comm, self.trace, self.plan
        S, s, q, m = self.S, self.s, self.q, self.m
        while True:
            R, Rl, off = self.R, self.Rl, self.off
            if R <= s:                                           # :60-63 nothing to reduce
                gids, mus = self._gather_survivors(S)
                keep = mus > 0
                return gids[keep], mus[keep]
            final = R <= S                                       # :65-74 single reduction of the points
            if plan.objective and not final:
                raise RuntimeError("recombination with an objective needs a pool of at most 2 * num_pts points: the "
                                   "reference fails here too (SOBER/_rchq.py:140-142 adds a [S, 1] sum in place to a "
                                   "[1, S] buffer)")
            S_r = R if final else S
            geo = RoundGeometry.of(R, S_r)
            t0 = time.perf_counter()
            retry = getattr(self, "_retry_msg", None)
            if retry is not None:                                # same round again (cluster time-out): reuse its message
                msg, Mc, C_cur, reg_blocks = retry
                self._retry_msg = None
            else:
                msg, Mc, C_cur, reg_blocks = self.sums.message(geo, S_r, final, pre)
                if plan.warp == "wsabim" and Mc is None:         # (a class round has added the term to its class messages)
                    # + U @ (0.5 sum mu cov^2): the one term of wsabim_kernel that is not linear in the block sums
                    with _Timer(ops, trace, "wsabim_sq"):
                        E = self.sums.wsabim_square_term(geo, S_r)
                        if plan.sober and not final and geo.n_tail > 0:
                            # SOBER/_rchq.py:127-135 counts the remainder's kernel columns a second time, in sets
                            # 0..N_rest-1: the whole kernel, hence its squared-covariance term too
                            E = E + self.sums.wsabim_square_term(geo, S_r, tail_as_block=True)
                        msg[1:q + 1] += _mm_splitk(ops, self.U, E, 8)
            msg0 = (msg, Mc, C_cur, reg_blocks)
            pre = None
            self.cls = None
            tail_row, n_tail_diag = 0, 0
            if self.diag_noise != 0.0 and not final and geo.n_tail > 0:
                # the ragged tail is a kernel block of its own (:91-99): predictive_covariance adds the noise to ITS
                # entries [k][k] too (tail point k x Nystrom row k).  One more message row carries the tail weights.
                tailw = ops.zeros(S_r)
                t0l = max(geo.n_full - off, 0)                   # first local tail position
                if t0l < Rl:
                    k0 = off + t0l - geo.n_full
                    tailw[k0:k0 + (Rl - t0l)] = self.mu[t0l:Rl] if self.wx is None else self.mu[t0l:Rl] * self.wx[t0l:Rl]
                msg = torch.cat([msg, tailw.unsqueeze(0)], 0)
                tail_row, n_tail_diag = msg.shape[0] - 1, min(m, geo.n_tail)
            self._trace_phase("blocksum+project", t0)
            t0 = time.perf_counter()
            if plan.objective:
                # SOBER/_rchq.py:78-104: one more feature per point, its objective (here still weighted by mu, like
                # every other message row); the reduction then keeps q + 2 points and the thinning removes one more
                return (yield from self._reduce_with_objective(msg))
            parts = comm.all_gather(msg) if comm.world > 1 else (msg if msg.dim() == 3 else msg.unsqueeze(0))
            M = S_r
            owner = self.owner
            replicate = cfg.REPLICATED_REDUCTION and comm.world > 1 and owner is None
            shared = comm.world > 1 and not replicate            # ONE rank reduces, the others receive the outcome
            red_rank = owner if owner is not None else 0
            XcarT = None
            cluster = not getattr(self, "_no_cluster", False)
            res, rv = ops.reduction_result(M) if shared else (None, None)
            if not shared or comm.rank == red_rank:
                ev_c = ops.record_event() if (trace is not None and trace.time_kernels and self._gpu_nullspace(M)) else None
                XcarT, tot = ops.finalize(parts, parts.shape[0], parts.shape[1], q, S_r, self.diagU, m, min(m, S_r),
                                          self.diag_noise, self.wrow, tail_row, n_tail_diag,
                                          tot_out=None if rv is None else rv["tot"])
                PhiT = yield from self._nullspace(XcarT, s, M, cluster)      # :140-143 (rows = null-space vectors)
                with _Timer(ops, trace, "eliminate"):
                    keep_rank, kept, w_star, info = ops.car_eliminate(PhiT, tot, M, s, cluster, out=rv)
                if ev_c is not None:
                    trace.chain_events.append((ev_c, ops.record_event()))
            elif not self._gpu_nullspace(M):
                yield _recorded_event(ops)                       # the reducing rank waits for its host SVD here: same yield count
            if shared:
                # one broadcast of the (tiny) reduction result: w_star | tot | info, kept, keep_rank
                comm.broadcast(res, src=red_rank)
                keep_rank, kept, w_star, info, tot = rv["keep_rank"], rv["kept"], rv["w_star"], rv["info"], rv["tot"]
            elif replicate:
                # a cluster-kernel time-out is local to one rank: the retry below must be a collective decision
                st = comm.all_gather(info[1:2].to(torch.float64))
                info[1:2] = st.max().to(torch.int32).reshape(1)
            Mn = None
            if Mc is not None and C_cur >= 2 and not final:
                # Enqueued BEFORE the host waits for this round's outcome: if exactly half of the sets survive (checked
                # below), the next round's class messages are a gather + rescale of this round's; otherwise the result
                # is dropped (the kernel tolerates a short survivor list).
                Mn = ops.empty(C_cur // 2 + self.sums.n_extra, Mc.shape[1], S_r)
                ops.regroup_classes(Mc[:C_cur], kept, w_star, tot, out=Mn[:C_cur // 2])
            head, ready = ops.to_host_async(ops.info_kept_buffer(info, kept), "head")   # one D2H: status + survivors
            yield ready
            hl = head.tolist()                                   # one conversion (iterating a tensor costs ~1 us/element)
            n_keep, status = hl[0], hl[1]
            kept_list = hl[2:2 + n_keep]
            if status == 2:
                # an 8-work-group cluster kernel gave up waiting for its siblings (they must be co-resident; a GPU shared
                # with other work may not grant that within the spin limit): nothing of this round has been applied yet --
                # redo its reduction, and every later one of the batch, on the single-work-group kernels
                if not cluster:
                    raise ReductionTimeout("a reduction kernel reported a time-out on the single-work-group path")
                self._no_cluster = True
                self.notes.append("a cluster reduction kernel timed out waiting for its sibling work-groups (GPU shared with "
                                  "other work?); the batch continued on the single-work-group kernels")
                self._retry_msg = msg0
                continue
            if status != 0 and not plan.sober:
                raise RuntimeError("Caratheodory elimination: a null vector has no positive entry "
                                   "(the reference fails here too: argmin of an empty tensor, _rchq.py:152)")
            if trace is not None:
                if trace.host_sync:
                    ops.synchronize()
                trace.add_time("reduce", time.perf_counter() - t0)
                rec = dict(R=R, S=S_r, nb=geo.nb, n_tail=geo.n_tail, kept=kept_list)
                if trace.keep_tensors:
                    rec["tot"] = tot.cpu()
                    if XcarT is not None:
                        rec["XcarT"] = XcarT.cpu()
                    rec["w_star"] = w_star[:n_keep].cpu()
                trace.rounds.append(rec)
            if final:
                gids, _ = self._gather_survivors(S)
                kept_t = torch.tensor(kept_list, dtype=torch.int64, device=gids.device)
                return gids[kept_t], w_star[:n_keep].clone()     # :69-73
            t0 = time.perf_counter()
            if Mc is not None and C_cur >= 2 and 2 * n_keep == S_r and status == 0:
                # exactly half of the sets survived: the next round's class messages are a gather + rescale of this round's
                if Mn is None:
                    Mn = ops.empty(C_cur // 2 + self.sums.n_extra, Mc.shape[1], S_r)
                    ops.regroup_classes(Mc[:C_cur], kept, w_star, tot, out=Mn[:C_cur // 2])
                self.cls = dict(M=Mn, C=C_cur // 2, reg_blocks=reg_blocks // 2)
            new_off, new_Rl = next_shard(off, Rl, geo, kept_list)
            self.cand, self.mu, self.gid, self.wx = ops.reweight_compact(
                self.cand, self.mu, self.gid, self.wx, Rl, off, geo.n_full, S_r, self.kp, keep_rank, w_star, tot, n_keep,
                new_off, new_Rl)
            self.R = survivors_before(R, geo, kept_list)
            self.off, self.Rl = new_off, new_Rl
            self.R_lo = min((self.R_lo // S_r) * s, self.R)
            self._trace_phase("compact", t0)

    def _gpu_nullspace(self, M):
        return cfg.GPU_NULLSPACE and M <= getattr(self.ops, "NULLSPACE_MAX_M", 1 << 30)

    def _nullspace(self, XcarT, s, M, cluster=True):
        """Rows s..M-1 of the full ``Vh`` of ``svd(XcarT)`` (:140-143; rows = null-space vectors)."""
        if self._gpu_nullspace(M):
            with _Timer(self.ops, self.trace, "nullspace"):
                return self.ops.nullspace(XcarT, s, M, cluster)
        return (yield from self._host_nullspace(XcarT, s, M))

    def _host_nullspace(self, XcarT, s, M):
        """The same rows from a full SVD on host LAPACK (``GPU_NULLSPACE = False``, or M beyond the kernels' limit)."""
        ops, trace = self.ops, self.trace
        if cfg.GPU_NULLSPACE and not getattr(self, "_warned_big_m", False):
            self._warned_big_m = True
            self.notes.append(f"2 * num_pts = {M} exceeds the GPU null-space kernels' limit "
                              f"({ops.NULLSPACE_MAX_M}): the per-round SVD runs on host LAPACK")
        t1 = time.perf_counter()
        Xh, ready = ops.to_host_async(XcarT, "xcar")
        yield ready
        with _lapack_threads(cfg.HOST_SVD_THREADS):
            Vh = torch.linalg.svd(Xh)[2]                         # :140 full SVD of [s, M] on host LAPACK
        PhiT = ops.from_host(Vh[-(M - s):, :], "phit")
        if trace is not None:
            trace.add_time("host_svd", time.perf_counter() - t1)
        return PhiT

    def _reduce_with_objective(self, msg):
        """Single reduction with an objective row (``SOBER/_rchq.py:77-111``), one process.

        ``msg`` = ``[tot ; U @ block sums]`` of the R points (one set each).  The Caratheodory step runs on
        ``[1 ; features ; objective]`` (q + 2 rows); then, among the kept points, the weights move along the null vector
        of ``[features ; 1]`` -- oriented so that the weighted objective does not decrease -- until one more reaches
        zero (``:87-104``).  That last step is k <= q + 2 numbers: host LAPACK, as in the reference.
        """
        ops, q, R, Rl = self.ops, self.q, self.R, self.Rl
        obj_row = (self.obj_live[:Rl] * self.mu[:Rl]).reshape(1, -1)
        parts = torch.cat([msg[:q + 1], obj_row], 0).unsqueeze(0).contiguous()
        XcarT, tot = ops.finalize(parts, 1, q + 2, q + 1, R, None, 0, 0, 0.0, 0)
        s_car = q + 2
        if R > s_car:
            PhiT = yield from self._nullspace(XcarT, s_car, R)
            _, kept, w_star, info = ops.car_eliminate(PhiT, tot, R, s_car)
            head, ready = ops.to_host_async(ops.info_kept_buffer(info, kept), "head")
            yield ready
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
        obj_p = self.obj_full.cpu()[kept_pos]                     # (sic) :89 indexes the objective by POSITION
        A = torch.cat([F, torch.ones(1, len(kept_pos), dtype=torch.float64)], 0)
        with _lapack_threads(cfg.HOST_SVD_THREADS):
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
        if self.trace is not None:
            self.trace.rounds.append(dict(R=R, S=R, nb=1, n_tail=0, kept=[int(v) for v in kept_pos]))
        gids = self.gid[:Rl]
        return gids[kept_pos.to(gids.device)], ops.to_device(w_host)

    def _gather_survivors(self, cap):
        """All ranks' (gid, mu) of the R <= cap survivors, in global position order, on every rank."""
        comm, ops = self.comm, self.ops
        gid, mu, Rl, R, off = self.gid, self.mu, self.Rl, self.R, self.off
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


# ----------------------------------------------------------------------------------------------------
# block-sum strategies
# ----------------------------------------------------------------------------------------------------
class FusedSums:
    """Structured kernels: the fused pairwise kernel (``basq_blocksum_f64``), per residue class where the plan allows."""

    def __init__(self, batch: Batch):
        self.b = batch
        self._geo_events = []                                   # (event pair, round, mode, info) awaiting the descriptor table
        # slots behind the C residue classes of an epoch: the irregular chunk (further blocks + the ragged remainder in set
        # S-1) and, for the SOBER variant, the remainder once more as a block of its own (point k in set k, no set weight:
        # SOBER/_rchq.py:127-135)
        self.n_extra = 2 if batch.plan.sober else 1

    def tail_block(self, geo_, S_, Xslot, totslot):
        """SOBER's first count of the remainder (host geometry) -> ``Xslot [1, m_ext, S]``; ``totslot [1, S]`` = 0."""
        b, ops = self.b, self.b.ops
        t0l = min(max(geo_.n_full - b.off, 0), b.Rl)            # first local remainder position
        ops.blocksum(b.spec, b.nys_ext, b.m_ext, b.cand[t0l:], b.mu[t0l:], None if b.wx is None else b.wx[t0l:], b.Rl - t0l,
                     b.off + t0l - geo_.n_full, S_, S_, 1, out=(Xslot, totslot))
        totslot.zero_()

    # -- launches (+ HIP events for the roofline line) -----------------------------------------------------
    def _timing(self):
        tr = self.b.trace
        return tr is not None and tr.time_kernels

    def timed(self, p_lo, p_hi, geo_, S_, n_ch, out, class_mod=0, class0=0):
        """One block-sum launch over the local positions [p_lo, p_hi)."""
        b, ops = self.b, self.b.ops
        ev0 = ops.record_event() if self._timing() else None
        clk = None
        if ev0 is not None and b.trace.sample_clock is not None and class_mod > 0:
            # one wave on a second stream, released by ev0: samples the clock every 250 us for the next 8 ms
            side = b.trace.sample_clock
            side.wait_event(ev0)
            clk = side.shader_clock_mhz(32, 250)
        ops.blocksum(b.spec, b.nys_ext, b.m_ext, b.cand[p_lo:], b.mu[p_lo:], None if b.wx is None else b.wx[p_lo:],
                     p_hi - p_lo, b.off + p_lo, geo_.n_full, S_, n_ch, out=out, class_mod=class_mod, class0=class0)
        if ev0 is not None and p_hi > p_lo:
            # pairs this launch evaluates: one class launch covers n_ch of class_mod classes of its range
            frac = (n_ch / class_mod) if class_mod else 1.0
            info = dict(pairs=float(p_hi - p_lo) * b.m_ext * frac, R=(p_hi - p_lo) * frac, m=b.m_ext, S=S_, chunks=n_ch,
                        class_mod=class_mod)
            b.trace.kernel_events.append((ev0, ops.record_event(), info))
            if clk is not None:
                info["clock_mhz"] = clk

    def timed_geo(self, r, mode, frac, launch):
        """A descriptor-driven launch; its pair count is filled in once the descriptor table has been read."""
        if not self._timing():
            launch()
            return
        ops = self.b.ops
        ev0 = ops.record_event()
        launch()
        self._geo_events.append((ev0, ops.record_event(), r, mode))

    def resolve_geo_events(self, table):
        b = self.b
        for ev0, ev1, r, mode in self._geo_events:
            R, reg_hi, off, Rl = (int(table[r][k]) for k in (0, 2, 6, 7))
            lo, hi = (0, reg_hi) if mode == 1 else ((reg_hi, R) if mode == 2 else (0, R))
            n = max(0, min(hi, off + Rl) - max(lo, off))
            if n > 0:
                b.trace.kernel_events.append((ev0, ev1, dict(pairs=float(n) * b.m_ext, R=n, m=b.m_ext, S=b.S, chunks=0)))
        self._geo_events = []

    # -- one round's block sums ------------------------------------------------------------------------------
    def irregular(self, geo_, S_, reg_blocks):
        """Block sums of the candidates the class partials do not cover (global positions >= reg_blocks * S: further
        blocks + the ragged tail), one chunk (SOBER: + the remainder as a block of its own) -> ``(Xirr [n_extra, m_ext, S], totirr [n_extra, S])``."""
        b, ops = self.b, self.b.ops
        Xirr, totirr = ops.empty(self.n_extra, b.m_ext, S_), ops.empty(self.n_extra, S_)
        reg_hi = min(max(reg_blocks * S_ - b.off, 0), b.Rl)             # local end of the regular region
        self.timed(reg_hi, b.Rl, geo_, S_, 1, (Xirr[:1], totirr[:1]))
        if self.n_extra == 2:
            self.tail_block(geo_, S_, Xirr[1:2], totirr[1:2])
        return Xirr, totirr

    def evaluate(self, geo_, S_, defer_last=False):
        """A fresh evaluation of one round's block sums -> ``(Xbuf [n, m_ext, S], totbuf [n, S], n, C, reg_blocks, late_fn)``.

        C >= 2: the regular region -- the first ``reg_blocks`` (a multiple of C) blocks -- is summed per residue
        class (slots 0..C-1), the rest (further blocks + ragged tail) is one contiguous chunk (slot C = n - 1).
        C == 1 (small rounds, variants without class sums): plain contiguous chunks.  ``defer_last``: the last chunk
        / class (and the irregular chunk) are returned as ``late_fn`` instead of being launched (round 1: they run
        behind the range finder's GPU work).  The class count follows the LOWER BOUND of the survivor count
        (``Batch.R_lo``), a function of N alone, so that every path -- descriptor-driven or not -- sums in one order."""
        b, ops = self.b, self.b.ops
        m_ext, off, Rl, kp = b.m_ext, b.off, b.Rl, b.kp
        C = classes_for(b.R_lo // S_) if (b.plan.classes and S_ == b.S) else 1
        n_late_chunks = 0 if b.pipelined else cfg.LATE_CHUNKS
        n_late_classes = cfg.LATE_CLASSES_PIPELINED if b.pipelined else cfg.LATE_CLASSES
        if C == 1:
            n_ch = choose_chunks(local_blocks(off, Rl, geo_), m_ext, S_, kp // 4)
            sober_tail = self.n_extra == 2 and S_ == b.S and geo_.n_tail > 0     # one more chunk: the remainder's first count
            n_tot = n_ch + (1 if sober_tail else 0)
            Xbuf, totbuf = ops.empty(n_tot, m_ext, S_), ops.empty(n_tot, S_)
            if sober_tail:
                self.tail_block(geo_, S_, Xbuf[n_ch:], totbuf[n_ch:])
            p_split = late_split(off, Rl, geo_.n_full, S_, n_ch, n_late_chunks) if (defer_last and Rl > 0) else None
            if p_split is None:
                self.timed(0, Rl, geo_, S_, n_ch, (Xbuf[:n_ch], totbuf[:n_ch]))
                return Xbuf, totbuf, n_tot, 1, 0, None
            # the last chunk(s) are launched behind the range finder's GPU work; same chunk boundaries, same sums
            self.timed(0, p_split, geo_, S_, n_ch - n_late_chunks, (Xbuf[:n_ch - n_late_chunks], totbuf[:n_ch - n_late_chunks]))
            return (Xbuf, totbuf, n_tot, 1, 0,
                    lambda: self.timed(p_split, Rl, geo_, S_, n_late_chunks, (Xbuf[n_ch - n_late_chunks:n_ch], totbuf[n_ch - n_late_chunks:n_ch])))
        reg_blocks = (geo_.nb // C) * C
        n_slots = C + self.n_extra
        Xbuf, totbuf = ops.empty(n_slots, m_ext, S_), ops.empty(n_slots, S_)
        reg_hi = min(max(reg_blocks * S_ - off, 0), Rl)                  # local end of the regular region

        def irregular():
            self.timed(reg_hi, Rl, geo_, S_, 1, (Xbuf[C:C + 1], totbuf[C:C + 1]))
            if self.n_extra == 2:
                self.tail_block(geo_, S_, Xbuf[C + 1:C + 2], totbuf[C + 1:C + 2])

        if defer_last and n_late_classes > 0:
            L = max(1, min(n_late_classes, C - 1))               # classes evaluated behind the range finder's GPU work
            self.timed(0, reg_hi, geo_, S_, C - L, (Xbuf[:C - L], totbuf[:C - L]), class_mod=C, class0=0)

            def late_fn():
                self.timed(0, reg_hi, geo_, S_, L, (Xbuf[C - L:C], totbuf[C - L:C]), class_mod=C, class0=C - L)
                irregular()

            return Xbuf, totbuf, n_slots, C, reg_blocks, late_fn
        self.timed(0, reg_hi, geo_, S_, C, (Xbuf[:C], totbuf[:C]), class_mod=C, class0=0)
        irregular()
        return Xbuf, totbuf, n_slots, C, reg_blocks, None

    def message(self, geo, S_r, final, pre):
        """-> ``(msg, Mc, C_cur, reg_blocks)``: the round's message ``[rows, S_r]`` -- or, on one rank without an extra
        message row, the class messages ``[C + 1, rows, S_r]`` as they are (the finalize kernel adds its parts in index
        order, exactly the sum a separate launch would have formed first)."""
        b, ops, trace, comm = self.b, self.b.ops, self.b.trace, self.b.comm
        sum_here = comm.world > 1 or (b.diag_noise != 0.0 and geo.n_tail > 0)
        if b.cls is not None and not final and S_r == b.S:
            # inside an epoch: the class messages were regrouped from the previous round's; only the candidates they
            # do not cover are evaluated (a few blocks + the ragged tail)
            Mc, C_cur, reg_blocks = b.cls["M"], b.cls["C"], b.cls["reg_blocks"]
            with _Timer(ops, trace, "blocksum"):
                Xirr, totirr = self.irregular(geo, S_r, reg_blocks)
            with _Timer(ops, trace, "project"):
                ops.project_chunks(b.U_ext, b.q_ext, b.m_ext, Xirr, totirr, self.n_extra, S_r, b.kscale,
                                   out=Mc[C_cur:C_cur + self.n_extra])
                if b.plan.warp == "wsabim":
                    with _Timer(ops, trace, "wsabim_sq"):
                        noise_part = self.wsabim_class_round(geo, S_r, Mc, C_cur, reg_blocks, fresh=False)
                    msg = ops.sum_parts(Mc)
                    if noise_part is not None:
                        msg += noise_part
                else:
                    msg = ops.sum_parts(Mc) if sum_here else Mc
            return msg, Mc, C_cur, reg_blocks
        with _Timer(ops, trace, "blocksum"):
            if pre is not None:
                Xpart, totpart, n_chunks, C_cur, reg_blocks = pre[:5]
            else:
                Xpart, totpart, n_chunks, C_cur, reg_blocks, _ = self.evaluate(geo, S_r)
        if C_cur >= 2:
            # start of an epoch: one message per residue class; the [m, S] partials are not needed again
            with _Timer(ops, trace, "project"):
                Mc = ops.project_chunks(b.U_ext, b.q_ext, b.m_ext, Xpart, totpart, n_chunks, S_r, b.kscale)
                if b.plan.warp == "wsabim":
                    with _Timer(ops, trace, "wsabim_sq"):
                        noise_part = self.wsabim_class_round(geo, S_r, Mc, C_cur, reg_blocks, fresh=True)
                    msg = ops.sum_parts(Mc)
                    if noise_part is not None:
                        msg += noise_part
                else:
                    msg = ops.sum_parts(Mc) if sum_here else Mc
            return msg, Mc, C_cur, reg_blocks
        # (SOBER/_rchq.py:127-135 -- the remainder's kernel columns also go to sets 0..N_rest-1, no weight added -- is one more
        # chunk of ``evaluate``'s result)
        with _Timer(ops, trace, "project"):
            msg = ops.project(b.U_ext, b.q_ext, b.m_ext, Xpart, totpart, n_chunks, S_r, b.kscale)
        return msg, None, 1, 0

    def _kobs_live(self):
        """``outputscale * k(Xobs, x_p)`` of this rank's live candidates -> ``[n_obs4, Rl]`` (rows beyond n_obs zero)."""
        b, ops = self.b, self.b.ops
        n4, Rl = b.bmatT.shape[0], max(b.Rl, 1)
        kobs = ops.empty(n4, Rl)
        if n4 != b.n_obs:
            kobs[b.n_obs:].zero_()                              # only the padding rows (the fragment loads read whole groups of 4)
        if b.Rl:
            ops.gram_into(b.spec, b.nys_ext[b.m:b.m + b.n_obs], b.n_obs, b.cand, b.Rl, kobs)   # rows m.. of nys_ext = packed observations
        return kobs

    def wsabim_class_round(self, geo, S, Mc, C_cur, reg_blocks, fresh):
        """WSABI-M's ``0.5 cov^2`` (``_wsabi.py:240-242``) in a round whose block sums are kept per residue class.

        ``0.5 (c + noise [j == kappa])^2 = 0.5 c^2 + [j == kappa] (noise c + 0.5 noise^2)``, c = the noise-free posterior
        covariance.  The first term is a per-pair block sum like the kernel itself: per class at the start of an epoch
        (``fresh``; ``basq_blocksum_sq_f64`` in class mode), projected and ADDED to the class messages ``Mc`` -- from then on it
        is regrouped with them, and only the candidates outside the regular region are evaluated again.  The bracket sits on
        ONE Nystrom row per candidate -- the row of its position inside its block, which changes every round -- so it is
        evaluated every round (``basq_cov_diag_f64``, one thread per candidate) -> the returned ``[rows, S]`` part of the
        message (None without noise)."""
        b, ops = self.b, self.b.ops
        m, q, Rl, off, n_obs = b.m, b.q, b.Rl, b.off, b.n_obs
        kobs = self._kobs_live()
        reg_hi = min(max(reg_blocks * S - off, 0), Rl)           # local end of the regular region
        n_slots = (C_cur if fresh else 0) + self.n_extra
        Epart = ops.empty(n_slots, m, S)
        k = 0
        if fresh:
            ops.blocksum_sq(b.spec, b.nys_ext, m, b.cand, b.mu, reg_hi, off, geo.n_full, S, C_cur, b.bmatT, kobs, n_obs, 0.0,
                            class_mod=C_cur, class0=0, out=Epart[:C_cur])
            k = C_cur
        ops.blocksum_sq(b.spec, b.nys_ext, m, b.cand[reg_hi:], b.mu[reg_hi:], Rl - reg_hi, off + reg_hi, geo.n_full, S, 1,
                        b.bmatT, kobs[:, reg_hi:], n_obs, 0.0, out=Epart[k:k + 1])
        t0l = min(max(geo.n_full - off, 0), Rl)                  # first local remainder position
        if self.n_extra == 2:                                    # SOBER's first count of the remainder: point k in set k
            ops.blocksum_sq(b.spec, b.nys_ext, m, b.cand[t0l:], b.mu[t0l:], Rl - t0l, off + t0l - geo.n_full, S, S, 1,
                            b.bmatT, kobs[:, t0l:], n_obs, 0.0, out=Epart[k + 1:k + 2])
        Me = ops.project_chunks(b.U, q, m, Epart, ops.zeros(n_slots, S), n_slots, S, 1.0)
        slots = Mc[:C_cur + self.n_extra] if fresh else Mc[C_cur:C_cur + self.n_extra]
        slots[:, 1:q + 1] += Me[:, 1:q + 1]
        if b.diag_noise == 0.0:
            return None
        val = ops.cov_diag(b.spec, b.nys_ext, m, b.cand, Rl, off, geo.n_full, S, b.bmatT, kobs, n_obs, b.diag_noise)
        part = ops.zeros(Mc.shape[1], S)
        if t0l > 0:                                              # full blocks: candidate in set s meets the noise on row s
            # dvec[s] = sum of mu_p val_p over the local candidates of set s: the shard's leading partial block, its whole
            # blocks as one [blocks, S] column sum, its trailing partial block (fixed shapes -> a fixed summation order)
            wv = b.mu[:t0l] * val[:t0l]
            dvec = ops.zeros(S)
            lead = min((-off) % S, t0l)
            if lead:
                dvec[off % S:off % S + lead] += wv[:lead]
            nbk = (t0l - lead) // S
            if nbk:
                dvec += wv[lead:lead + nbk * S].view(nbk, S).sum(0)
            if t0l - lead - nbk * S:
                dvec[:t0l - lead - nbk * S] += wv[lead + nbk * S:]
            nd = min(m, S)
            part[1:q + 1, :nd] = b.U[:, :nd] * dvec[:nd]
        if Rl > t0l:                                             # remainder: point k meets it on row k; all of it is in set S-1
            k0 = off + t0l - geo.n_full
            k1 = min(k0 + (Rl - t0l), m)
            if k1 > k0:
                dt = b.mu[t0l:t0l + (k1 - k0)] * val[t0l:t0l + (k1 - k0)]
                part[1:q + 1, S - 1] += b.U[:, k0:k1] @ dt
                if self.n_extra == 2:                            # ... and, SOBER, once more in set k
                    part[1:q + 1, k0:k1] += b.U[:, k0:k1] * dt
        return part

    def wsabim_square_term(self, geo, S, tail_as_block=False):
        """E[j, s] = 0.5 * sum_{p in set s} mu_p * cov(pt_j, x_p)^2  with cov = k - K(pt,X) W K(X, x)  (_wsabi.py:240).

        ``tail_as_block``: only the ragged remainder, as a kernel block of its own -- remainder point k in set k (the first
        of SOBER's two counts of the remainder, ``SOBER/_rchq.py:127-135``).

        ``cov`` is ``predictive_covariance``, which carries the likelihood noise on entry [k][k] of every block the
        reference builds: candidate p of a full block meets Nystrom row ``p mod S``, tail point k meets row k.

        Fused: one Gram launch for ``K(X, x_p)`` of the live candidates ([n_obs, Rl], the only per-candidate array),
        then ``basq_blocksum_sq_f64`` evaluates k, subtracts the correction (a second MFMA chain over the observations),
        squares and accumulates in registers -- no [m, candidates] covariance block exists.
        """
        b, ops = self.b, self.b.ops
        m, n_obs, Rl = b.m, b.n_obs, b.Rl
        cand, mu, off, n_full = b.cand, b.mu, b.off, geo.n_full
        if tail_as_block:
            t0l = min(max(geo.n_full - b.off, 0), Rl)            # first local tail position
            cand, mu, Rl = cand[t0l:], mu[t0l:], Rl - t0l
            off, n_full = b.off + t0l - geo.n_full, S            # positions renumbered from the start of the remainder
        if Rl == 0:
            return ops.zeros(m, S)
        n4 = b.bmatT.shape[0]
        kobs = ops.empty(n4, Rl)
        if n4 != n_obs:
            kobs[n_obs:].zero_()
        ops.gram_into(b.spec, b.nys_ext[m:m + n_obs], n_obs, cand, Rl, kobs)   # rows m.. of nys_ext = packed observations
        n_ch = 1 if tail_as_block else choose_chunks(local_blocks(off, Rl, geo), m, S, b.kp // 4)
        Epart = ops.blocksum_sq(b.spec, b.nys_ext, m, cand, mu, Rl, off, n_full, S, n_ch, b.bmatT, kobs, n_obs,
                                b.diag_noise)
        return Epart[0] if n_ch == 1 else ops.sum_parts(Epart)


class OpaqueSums:
    """An opaque callable (the reference's own ``kernel`` contract): no packing, no fused kernel -- the candidates stay raw
    ``[R, d]`` rows and every round's block sums come from dense kernel blocks through ``basq_dense_blocksum_f64``."""

    def __init__(self, batch: Batch):
        self.b = batch

    def message(self, geo, S_r, final, pre):
        b, ops = self.b, self.b.ops
        with _Timer(ops, b.trace, "blocksum"):
            Xpart, totpart = self.block_sums(geo.n_full, S_r)
        with _Timer(ops, b.trace, "project"):
            msg = ops.project(b.U_ext, b.q_ext, b.m_ext, Xpart, totpart, 1, S_r, b.kscale)
        return msg, None, 1, 0

    def block_sums(self, n_full, S):
        """``X_for_nys`` and ``tot_weights`` of ``_rchq.py:79-99`` as ``(Xpart [1, m, S], totpart [1, S])``, same layout as
        ``basq_blocksum_f64`` with one chunk.

        ``block_exact`` mode (the default whenever the callable's value depends on the block it is asked for -- decided
        by ``CallableKernel.resolve_mode``'s probe, e.g. ``predictive_covariance``'s per-block noise diagonal): the
        reference's own calls, one ``kernel(pts_nys, block)`` per block of S points (``:81-86``) and one for the ragged
        tail (``:91-99``).  On several ranks a block that straddles a shard border is evaluated, whole, by the rank that
        owns its FIRST point, which borrows the missing points from its successors (``_borrow``).

        Chunked mode: ``C = kernel(pts_nys, chunk)`` ([m, nc] float64 on the device, at most ``chunk_bytes``) per chunk
        of consecutive candidates, summed into the sets by ``basq_dense_blocksum_f64`` in position order (the set
        weights through the same kernel with an all-ones row)."""
        b, ops, kernel = self.b, self.b.ops, self.b.kernel
        m, Rl, off, R = b.m, b.Rl, b.off, b.R
        E, T = ops.zeros(m, S), ops.zeros(1, S)
        if b.exact_blocks:
            first, need = exact_unit_plan(off, Rl, n_full, R, S)
            cand, mu = b.cand[:Rl], b.mu[:Rl]
            if b.comm.world > 1:
                cand, mu = self._borrow(cand, mu, need, S)
            p = (first - off) if first is not None else Rl        # local index of the first unit this rank evaluates
            while p < Rl:
                pg = off + p
                hi = p + S if pg < n_full else R - off            # a block (:81-86) or the remainder (:91-99)
                Kb = kernel.dense(ops, b.pts_nys, cand[p:hi])
                ops.dense_blocksum(Kb, mu[p:hi], pg, n_full, S, 1.0, E)
                p = hi
            if Rl > 0:
                ones = ops.zeros(1, Rl) + 1.0
                ops.dense_blocksum(ones, b.mu[:Rl], off, n_full, S, 1.0, T)
            return E.unsqueeze(0), T
        if Rl == 0:
            return E.unsqueeze(0), T
        nc_max = max(S, min(Rl, kernel.chunk_bytes // (8 * m)))
        nc_max = (nc_max // S) * S                              # whole blocks: every chunk starts at the same set, and at an
                                                                # even position when the shard does (16-byte loads, see the kernel)
        for p0 in range(0, Rl, nc_max):
            nc = min(nc_max, Rl - p0)
            Kc = kernel.dense(ops, b.pts_nys, b.cand[p0:p0 + nc])
            ops.dense_blocksum(Kc, b.mu[p0:p0 + nc], off + p0, n_full, S, 1.0, E, tot=T)   # set weights in the same launch
        return E.unsqueeze(0), T

    def _borrow(self, cand, mu, need, S):
        """Multi-rank ``block_exact``: append the ``need`` candidates that follow this rank's shard (``need < S``).

        Every rank publishes its first ``S - 1`` live candidates and their weights (ONE all-gather of ``[S, d + 1]`` rows
        per round); a rank whose last block (or the ragged tail) runs past its shard takes the missing points from its
        successors' heads, in rank order."""
        b, ops, comm = self.b, self.b.ops, self.b.comm
        d, Rl = b.d, b.Rl
        H = S - 1
        head = ops.zeros(H + 1, d + 1)
        nh = min(H, Rl)
        head[0, 0] = float(Rl)
        if nh:
            head[1:1 + nh, :d] = b.cand[:nh]
            head[1:1 + nh, d] = b.mu[:nh]
        allh = comm.all_gather(head)                             # [W, S, d + 1]
        if need <= 0:
            return cand, mu
        counts = [int(v) for v in allh[:, 0, 0].cpu()]
        extra_c, extra_m = [], []
        for r in range(comm.rank + 1, comm.world):
            take = min(counts[r], need, H)
            if take > 0:
                extra_c.append(allh[r, 1:1 + take, :d])
                extra_m.append(allh[r, 1:1 + take, d])
                need -= take
            if need <= 0:
                break
        assert need <= 0, "successor shards do not cover the straddling block"
        return torch.cat([cand] + extra_c, 0).contiguous(), torch.cat([mu] + extra_m, 0).contiguous()


def exact_unit_plan(off: int, Rl: int, n_full: int, R: int, S: int):
    """Which of the reference's kernel calls (blocks of S positions below ``n_full``, then ONE call for the remainder
    ``[n_full, R)``) the rank holding positions ``[off, off + Rl)`` makes: those whose FIRST position it holds.
    -> ``(first, need)``: the global position of its first call (None: it makes none) and how many positions beyond its
    shard its last call reaches (< S)."""
    end = off + Rl
    if Rl == 0:
        return None, 0
    if off <= n_full:
        first = min(-(-off // S) * S, n_full)
    else:
        return None, 0                                           # inside the remainder, which a predecessor owns
    if first >= end or first >= R:
        return None, 0
    last_start = ((end - 1) // S) * S if (end - 1) < n_full else n_full
    unit_end = last_start + S if last_start < n_full else R
    return first, max(0, unit_end - end)
