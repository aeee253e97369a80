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

Modules (round 6: one 1 300-line module split, no behaviour change): this file = the batch's state, set-up, basis and step
generator; ``_plan.py`` = ``Plan`` + shared helpers; ``_sums.py`` = the block-sum strategies; ``_epochs.py`` = descriptor-driven
rounds with the irregular candidates as message columns (one rank, BASQ variant, no WSABI-M); ``_rounds_async.py`` = the
descriptor-driven rounds of everything else; ``_rounds_sync.py`` = rounds with one read-back each + the reduction's host pieces.

The batch never blocks: wherever the host has to wait for the GPU, ``steps()`` yields the event, and the engine decides
whether to block on it (one batch) or to advance another batch meanwhile (``RecombinationEngine.run_many``).
"""
from __future__ import annotations

import dataclasses
import time
import warnings

import torch

from . import _config as cfg
from . import _epochs
from ._basis import _mm_splitk, _ShardedProducts, _skip_test_matrix_draw, _Timer, make_cov_psd, nystrom_basis_steps
from ._lib import ROLE_A, ROLE_B
from ._partition import RoundGeometry, initial_shards
from ._plan import Plan, ReductionTimeout, _NoWait, _recorded_event, classes_for, late_split      # noqa: F401  (re-exported)
from ._rounds_async import AsyncRounds
from ._rounds_sync import SyncRounds
from ._sums import FusedSums, OpaqueSums, exact_unit_plan                                        # noqa: F401  (re-exported)


class Batch(AsyncRounds, SyncRounds):
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
