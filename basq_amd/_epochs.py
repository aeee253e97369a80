"""Descriptor-driven rounds with NO pairwise evaluation inside an epoch (round 6; ``BASQ/_rchq.py:76-130``).

``AsyncRounds._async_rounds`` regroups the residue-class messages from round to round but still evaluates, projects and compacts
the candidates outside the classes -- the ``e < C`` full blocks behind the regular region and the ragged tail -- in EVERY round:
five chip-wide launches (60-70 us) between two chains of single-work-group kernels.  Those candidates obey the same law as the
classes: a round sends the survivor of (block b, kept rank k) to position ``b * n_keep + k`` with its weight rescaled by
``w*_k / tot``, the tail -- if set S-1 survives -- follows behind, and the regular survivors fill the next regular region
exactly.  Kept as MESSAGE COLUMNS (one ``[rows]`` column per candidate: slot b = block b, one more slot = the tail), they form a
closed system that ``basq_epoch_turn_f64`` advances together with the classes and the round descriptor: ONE launch between an
elimination and the next round's finalize.  No candidate is touched inside an epoch, so the rounds' compactions are applied
together when the next epoch (or the host's round-by-round loop) needs the candidates again (``basq_reweight_compact_rounds_f64``).

What the columns cost -- one block-sum chunk and one projection per irregular block instead of one for all of them -- is paid
BESIDE the epoch's first chain, on the device's side stream: that chain keeps one compute unit busy for ~0.4 ms and needs none
of it; the launch stream waits for the side stream only before the first ``epoch_turn``.

Scope: BASQ variant, stationary / posterior / WSABI-L kernels (``predictive_covariance``'s likelihood noise on the block diagonals
included: the tail block's per-point weights, which that term needs, are a row of the tail slot); one rank, or several (replicated
or owner-rank reductions) -- every rank then carries the PARTIAL class messages and columns of its shard (``epoch_turn`` is linear in
them: a gather + rescale), the summed message is all-gathered once per round exactly as before, and the compaction of an
epoch's rounds walks the rank's shard through the descriptors.  WSABI-M and the SOBER variant keep ``AsyncRounds._async_rounds``.
"""
from __future__ import annotations

import contextlib

import torch

from . import _config as cfg
from ._partition import choose_chunks
from ._plan import classes_for


def eligible(b) -> bool:
    """Can batch ``b`` (operands prepared) take the column form of the descriptor-driven rounds?"""
    plan = b.plan
    return bool(cfg.IRR_COLUMNS and plan.async_rounds and plan.classes
                and (b.comm.world == 1 or cfg.REPLICATED_REDUCTION or b.owner is not None)
                and not plan.sober and plan.warp != "wsabim" and hasattr(b.ops, "epoch_turn"))


def block_capacity(R_lo: int, R_up: int, S: int, C: int) -> int:
    """Upper bound of the number of full blocks behind the regular region (``nb mod C``) over the possible block counts."""
    return max(nb % C for nb in range(R_lo // S, R_up // S + 1))


def async_rounds_columns(b, pre):
    """Generator with the contract of ``AsyncRounds._async_rounds`` (-> False | True = a round violated the plan | "basis")."""
    ops, trace, comm = b.ops, b.trace, b.comm
    owner = b.owner                                              # None: every rank reduces; else: that rank + a broadcast
    multi = comm.world > 1                                       # several ranks: this rank's shard of the candidates and its PARTIAL
    S, s, q, m_ext, q_ext = b.S, b.s, b.q, b.m_ext, b.q_ext     # messages / columns (epoch_turn is linear); one all-gather per round
    spec, nys_ext, U_ext, kscale, kp = b.spec, b.nys_ext, b.U_ext, b.kscale, b.kp
    n = s                                                        # a regular round keeps s = S / 2 sets
    rows = q_ext + 1
    reg_hi0 = (pre[4] * S) if (pre is not None and pre[3] >= 2) else 0
    geo_t = ops.geo_init(64, b.R, S, reg_hi0, b.off, b.Rl)
    r = 0
    R_lo = R_up = b.R
    Rl_up = b.Rl                                                 # upper bound of this rank's shard (sizes launches / buffers)
    cand, mu, gid, wx = b.cand, b.mu, b.gid, b.wx
    pend, pend_r0, pend_R = [], 0, b.Rl                          # rounds whose compaction is still owed; first row; bound of len(cand)
    P, C_cur, E_cur = None, 1, 0                                 # this round's buffer [C | fold | E | tail] while inside an epoch
    plan_C = None
    records = []
    side_ev, side_keep = None, None
    side_rounds = []                                             # rounds whose columns were evaluated (for the trace)

    def flush(out_rows):
        """Apply the pending rounds' compactions (one launch) -> candidates of round ``r``."""
        nonlocal cand, mu, gid, wx, pend
        if pend:
            cand, mu, gid, wx = ops.reweight_compact_rounds(cand, mu, gid, wx, geo_t[pend_r0:], pend, pend_R, S, kp, out_rows, n)
            pend = []

    while R_lo > S:
        g_row = geo_t[r]
        fresh = P is None
        if fresh:
            # ---- a fresh evaluation: the start of an epoch (C >= 2), or a round without classes
            flush(Rl_up)
            pend_r0, pend_R = r, Rl_up
            if pre is not None:                                  # round 1: launched before the basis, host geometry
                Xpart, totpart, n_chunks, C_cur = pre[:4]
                pre = None
            else:
                C_cur = plan_C if plan_C is not None else 1
                if C_cur >= 2:
                    n_chunks = C_cur + 1
                    Xpart, totpart = ops.empty(n_chunks, m_ext, S), ops.empty(n_chunks, S)
                    b.sums.timed_geo(r, 1, 1.0, lambda: ops.blocksum_geo(
                        spec, nys_ext, m_ext, cand, mu, wx, g_row, 1, S, C_cur, out=(Xpart[:C_cur], totpart[:C_cur]),
                        class_mod=C_cur))
                    b.sums.timed_geo(r, 2, 1.0, lambda: ops.blocksum_geo(
                        spec, nys_ext, m_ext, cand, mu, wx, g_row, 2, S, 1, out=(Xpart[C_cur:], totpart[C_cur:])))
                else:
                    n_chunks = choose_chunks(max(R_lo // S // comm.world, 1), m_ext, S, kp // 4)
                    Xpart, totpart = ops.empty(n_chunks, m_ext, S), ops.empty(n_chunks, S)
                    b.sums.timed_geo(r, 3, 1.0, lambda: ops.blocksum_geo(spec, nys_ext, m_ext, cand, mu, wx, g_row, 3, S,
                                                                         n_chunks, out=(Xpart, totpart)))
            if C_cur >= 2:
                E_cur = block_capacity(R_lo, R_up, S, C_cur)
                P = ops.empty(C_cur + E_cur + 2, rows, S)
                # classes + the ordinary irregular chunk, whose message IS the fold slot of this round
                ops.project_chunks(U_ext, q_ext, m_ext, Xpart, totpart, C_cur + 1, S, kscale, out=P[:C_cur + 1])
                # ... and the columns the NEXT rounds of the epoch regroup, beside this round's chain (side stream)
                # (with other batches in flight there is no idle chip to fill, and a stream shared between the batches only ties
                #  them together: the same launches go to the batch's own stream then -- same arithmetic, same bits)
                own = b.pipelined and cfg.PIPELINED_SIDE_STREAM and hasattr(ops, "own_side_ops")
                side = ops.own_side_ops() if own else (ops if b.pipelined else ops.side_ops())
                if side is not ops:
                    side.wait_event(ops.record_event(False))
                with (contextlib.nullcontext() if side is ops else (ops.own_side_context() if own else ops.side_context())):
                    Xirr, totirr = side.empty(E_cur + 1, m_ext, S), side.empty(E_cur + 1, S)
                    timing = b.sums._timing()                    # (HIP events on the stream these launches run on)
                    if E_cur > 0:
                        ev0 = side.record_event() if timing else None
                        side.blocksum_geo(spec, nys_ext, m_ext, cand, mu, wx, g_row, 5, S, E_cur, out=(Xirr[:E_cur], totirr[:E_cur]),
                                          class_mod=C_cur)
                        if timing:
                            b.sums._geo_events.append((ev0, side.record_event(), r, 5))
                    ev0 = side.record_event() if timing else None
                    side.blocksum_geo(spec, nys_ext, m_ext, cand, mu, wx, g_row, 4, S, 1, out=(Xirr[E_cur:], totirr[E_cur:]))
                    if timing:
                        b.sums._geo_events.append((ev0, side.record_event(), r, 4))
                    side.project_chunks(U_ext, q_ext, m_ext, Xirr, totirr, E_cur + 1, S, kscale, out=P[C_cur + 1:])
                    if side is not ops:
                        side_ev = side.record_event(False)
                        side_keep = (Xirr, totirr)
                side_rounds.append(r)
                parts = P[:C_cur + 1]
            else:
                parts = ops.project(U_ext, q_ext, m_ext, Xpart, totpart, n_chunks, S, kscale).unsqueeze(0)
            del Xpart, totpart
        else:
            parts = P[:C_cur + 1]
        if b.diag_noise != 0.0:
            # predictive_covariance's noise on the ragged tail block (BASQ/_gp.py:275-276: entries [k][k], tail point k x Nystrom
            # row k): one more message row carries the tail points' weights -- from the candidates where the round evaluated
            # them, from the tail slot's columns (set weight of a one-point set; the kernel-weighted one for WSABI-L) inside an epoch
            buf = ops.empty(1, rows + 1, S)
            ops.sum_parts(parts, out=buf[0, :rows])
            if fresh:
                ops.tail_weights_geo(mu, wx, g_row, S, buf[0, rows])
            else:
                buf[0, rows].copy_(P[C_cur + 1 + E_cur][b.wrow])
            if multi:
                buf = comm.all_gather(buf[0])                    # [world, rows + 1, S], added in rank order by the finalize kernel
            fin = (buf, buf.shape[0], rows + 1, q, S, b.diagU, b.m, min(b.m, S), b.diag_noise, b.wrow, rows, min(b.m, S), g_row)
        else:
            if multi:
                parts = comm.all_gather(ops.sum_parts(parts) if parts.shape[0] > 1 else parts[0])
            fin = (parts, parts.shape[0], rows, q, S, None, b.m, min(b.m, S), 0.0, 0, 0, 0, None)
        # ---- the round's chain of single-work-group kernels
        # (owner-rank mode -- batches in flight on several ranks: the chain runs on ONE rank, its outcome, 3 S + 1 doubles, is
        #  broadcast stream-ordered on the batch's own process group; every rank then advances its own partial messages with it)
        res, rv = ops.reduction_result(S) if owner is not None else (None, None)
        if owner is None or comm.rank == owner:
            ev_c = ops.record_event() if b.sums._timing() else None
            XcarT, tot = ops.finalize(*fin, tot_out=None if rv is None else rv["tot"])
            PhiT = ops.nullspace(XcarT, s, S)
            keep_rank, kept, w_star, info = ops.car_eliminate(PhiT, tot, S, s, out=rv)
            if ev_c is not None:
                trace.chain_events.append((ev_c, ops.record_event()))
        if owner is not None:
            comm.broadcast(res, src=owner)
            keep_rank, kept, w_star, info, tot = rv["keep_rank"], rv["kept"], rv["w_star"], rv["info"], rv["tot"]
        if trace is not None:
            records.append(ops.info_kept_buffer(info, kept))
        pend.append(dict(keep_rank=keep_rank, w_star=w_star, tot=tot, info=info))
        # bounds of the next survivor count; the class plan of the next fresh evaluation follows the lower one
        R_lo_n = (R_lo // S) * n
        R_up_n = (R_up // S) * n + (S - 1)
        # this rank's shard [off, off + Rl): at most ceil(Rl / S) + 1 blocks touch it, each keeps n; + the tail
        Rl_up_n = min(R_up_n, (-(-Rl_up // S) + 1) * n + (S - 1)) if multi else R_up_n
        if P is not None and C_cur >= 2:
            # next round's classes, columns, fold slot and descriptor: one launch
            if side_ev is not None:
                ops.wait_event(side_ev)                          # (the columns of the epoch's first round)
                side_ev, side_keep = None, None
            E_next = (E_cur * n + S - 1) // S
            P = ops.epoch_turn(P, C_cur, E_cur, E_next, kept, keep_rank, w_star, tot, info, g_row, geo_t[r + 1])
            C_cur, E_cur = C_cur // 2, E_next
            plan_C = None
        else:
            P, C_cur, E_cur = None, 1, 0                         # the epoch is over (or there was none): next round evaluates afresh
            plan_C = classes_for(R_lo_n // S) if b.plan.classes else 1
            ops.round_next(g_row, info, keep_rank, S, plan_C if plan_C >= 2 else 0, True, geo_t[r + 1])
        r += 1
        R_lo, R_up, Rl_up = R_lo_n, R_up_n, Rl_up_n
        if len(pend) >= 8:                                       # (never with C <= 16: an epoch has at most five rounds)
            flush(Rl_up)
            pend_r0, pend_R = r, Rl_up
    flush(Rl_up)                                                 # the host's loop needs the candidates
    if side_ev is not None:                                      # (the last enqueued round opened an epoch: its columns are not
        ops.wait_event(side_ev)                                  #  used, but their buffer must outlive the side stream's writes)
        side_ev, side_keep = None, None
    if multi and owner is None:
        # every rank ran its own reductions: a cluster-kernel time-out (status 2) is local to ONE rank, and the ranks must agree on
        # repeating the rounds -- the flag becomes the maximum over the ranks
        flags = comm.all_gather(geo_t[r, 3:4].to(torch.float64))
        geo_t[r, 3:4] = flags.max().to(torch.int64).reshape(1)
    bad64 = (b._basis_bad != 0).to(torch.int64) if b._basis_bad is not None else geo_t[0, 3:4] * 0
    flat, ready = ops.to_host_async(torch.cat([geo_t[:r + 1].reshape(-1), bad64.reshape(1)]), "geo_table")
    yield ready                                                  # the ONE wait of the asynchronous rounds
    table = flat[:-1].view(r + 1, 8)
    if b._basis_bad is not None:
        b._basis_bad = None
        if int(flat[-1]) != 0:
            return "basis"
    row = table[r].tolist()
    if row[3] != 0:
        return True
    if trace is not None:
        b._trace_async_rounds(table, records, r)
        # the columns' kernel values (every candidate behind the regular region once more, per epoch)
        trace.side_pairs += float(sum(int(table[k][0]) - int(table[k][2]) for k in side_rounds)) * m_ext
    b.cand, b.mu, b.gid, b.wx = cand, mu, gid, wx
    b.R, b.off, b.Rl = int(row[0]), int(row[6]), int(row[7])
    b.R_lo = R_lo
    b.cls = None
    if P is not None:
        # mid-epoch hand-over: the host's loop evaluates the irregular candidates itself (slot C of its class messages)
        b.cls = dict(M=P[:C_cur + 1], C=C_cur, reg_blocks=int(row[2]) // S)
    return False
