"""What a batch runs, decided once (``Plan.of``), and the small helpers every part of the batch shares.

Split out of ``_batch.py`` in round 6 (no behaviour change): ``_batch.py`` keeps the batch's state, set-up and step generator;
``_sums.py`` the block-sum strategies; ``_rounds_async.py`` / ``_epochs.py`` the rounds enqueued without a host wait;
``_rounds_sync.py`` the rounds with one read-back each.  The SOBER variant (``SOBER/_rchq.py``) is not a module of its own: it is
the ``Plan.sober`` flag -- importance weights and ``make_cov_psd`` in ``_batch.py``, the remainder's second count
(``FusedSums.n_extra``, ``tail_block``) in ``_sums.py``, the early-stopping elimination in ``_rounds_sync.py``.
"""
from __future__ import annotations

from dataclasses import dataclass

from . import _config as cfg


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
