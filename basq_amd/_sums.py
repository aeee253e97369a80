"""Block-sum strategies: how a round's ``(q+1) x S`` message is formed (``BASQ/_rchq.py:79-101``).

* ``FusedSums``  -- structured kernels (``basq_amd.kernels``): the fused pairwise kernel (``basq_blocksum_f64``), per residue class
  of the block index where the round structure allows it, with the SOBER remainder columns (``tail_block``) and WSABI-M's
  squared-covariance term (``wsabim_class_round`` / ``wsabim_square_term``) as explicit add-ons;
* ``OpaqueSums`` -- any callable: dense chunks (or the reference's own block-by-block calls) through ``basq_dense_blocksum_f64``.

Split out of ``_batch.py`` in round 6 (no behaviour change).
"""
from __future__ import annotations

import torch

from . import _config as cfg
from ._basis import _Timer
from ._partition import choose_chunks, local_blocks
from ._plan import classes_for, late_split


# ----------------------------------------------------------------------------------------------------
# block-sum strategies
# ----------------------------------------------------------------------------------------------------
class FusedSums:
    """Structured kernels: the fused pairwise kernel (``basq_blocksum_f64``), per residue class where the plan allows."""

    def __init__(self, batch):
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
            R, n_full, reg_hi, off, Rl = (int(table[r][k]) for k in (0, 1, 2, 6, 7))
            lo, hi = {1: (0, reg_hi), 2: (reg_hi, R), 4: (n_full, R), 5: (reg_hi, n_full)}.get(mode, (0, R))
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

    def __init__(self, batch):
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
