"""The rounds that are certainly not the final one, enqueued without a host wait (device-resident round descriptor); the form
that evaluates the candidates outside the residue classes in every round -- several ranks, WSABI-M, the SOBER variant.  One rank
with a BASQ-variant stationary / posterior / WSABI-L kernel takes ``_epochs.async_rounds_columns`` instead.

Split out of ``_batch.py`` in round 6 (no behaviour change): a mix-in of ``Batch``.
"""
from __future__ import annotations

import torch

from ._basis import _mm_splitk
from ._partition import choose_chunks
from ._plan import classes_for


class AsyncRounds:
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
