"""The rounds with one read-back each: the last two or three of a batch, and every round of a batch that is traced phase by
phase, reduces wider than the kernels hold, uses an opaque callable or the SOBER objective row.  Also the per-round reduction's
host-side pieces (LAPACK null space, the objective row, the final gather).

Split out of ``_batch.py`` in round 6 (no behaviour change): a mix-in of ``Batch``.
"""
from __future__ import annotations

import time

import torch

from . import _config as cfg
from ._basis import _lapack_threads, _mm_splitk, _Timer
from ._partition import RoundGeometry, next_shard, survivors_before
from ._plan import ReductionTimeout, _recorded_event


class SyncRounds:
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
                # (the kept sets are on the device already: no host-to-device copy of the list the host has just read back --
                #  a synchronous 30-us copy at the very end of every batch, with the GPU idle)
                kept_t = kept[:n_keep].to(torch.int64) if kept.device == gids.device else \
                    torch.tensor(kept_list, dtype=torch.int64, device=gids.device)
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
