"""TEST-ONLY stand-in for ``basq_amd._ops.HipOps`` on the CPU.

It exists so the ``-m "not gpu"`` suite can exercise the *host logic* of
``basq_amd._engine`` (round geometry, sharding, offsets, collectives over gloo,
extended operands for the posterior / WSABI kernels) where no GPU is present.
It is never imported by the product (``basq_amd`` has no CPU path) and it says
nothing about the HIP kernels: those are checked by the ``-m gpu`` tests.

Each method restates, with plain torch ops, the contract documented for the
corresponding entry point in ``include/basq_hip.h``.
"""
from __future__ import annotations

import math

import numpy as np
import torch

ROLE_A, ROLE_B = 0, 1


def _dlarfg(alpha, x):
    """LAPACK dlarfg: H = I - tau [1; v][1; v]^T maps (alpha, x) to (beta, 0).  -> (tau, v)."""
    ss = float(np.dot(x, x))
    if ss == 0.0:
        return 0.0, np.zeros_like(x)
    beta = -math.copysign(math.sqrt(alpha * alpha + ss), alpha)
    return (beta - alpha) / beta, x / (alpha - beta)


def householder_nullspace(X):
    """Rows m..n-1 of the full ``Vh`` of ``torch.linalg.svd(X)`` for wide X [m, n], with no SVD iteration.

    gesdd reduces X to lower-bidiagonal form with Householder reflectors (dgebd2 order: right reflector from
    row i, applied to the rows below; left reflector from column i, applied to the trailing block); the
    rotations that follow only mix the first m rows of P^T, so the null-space rows are rows m.. of
    (G_0 ... G_{m-1})^T -- signs included.  Same algorithm as ``basq_nullspace_f64`` (host restatement, tests only).
    """
    A = X.detach().cpu().to(torch.float64).numpy().copy() if torch.is_tensor(X) else np.array(X, dtype=np.float64)
    m, n = A.shape
    taus = np.zeros(m)
    for i in range(m):
        taus[i], v = _dlarfg(A[i, i], A[i, i + 1:])
        A[i, i + 1:] = v
        if i < m - 1:
            vf = np.concatenate([[1.0], v])
            sub = A[i + 1:, i:]
            sub -= taus[i] * np.outer(sub @ vf, vf)
            tq, u = _dlarfg(A[i + 1, i], A[i + 2:, i])
            uf = np.concatenate([[1.0], u])
            sub = A[i + 1:, i + 1:]
            sub -= tq * np.outer(uf, uf @ sub)
    N = np.zeros((n, n - m))
    N[m:, :] = np.eye(n - m)
    for i in range(m - 1, -1, -1):
        v = np.zeros(n)
        v[i] = 1.0
        v[i + 1:] = A[i, i + 1:]
        N -= taus[i] * np.outer(v, v @ N)
    return torch.from_numpy(np.ascontiguousarray(N.T))


class _DoneEvent:
    def synchronize(self):
        pass


class CpuStandInOps:
    name = "cpu-stand-in"

    def __init__(self):
        self.device = torch.device("cpu")
        self.calls = {}

    def _count(self, k):
        self.calls[k] = self.calls.get(k, 0) + 1

    def empty(self, *shape, dtype=torch.float64):
        return torch.empty(*shape, dtype=dtype)

    def zeros(self, *shape, dtype=torch.float64):
        return torch.zeros(*shape, dtype=dtype)

    def to_device(self, t, dtype=None):
        return t.to(dtype=dtype or t.dtype).contiguous()

    def kp(self, d):
        return ((d + 2 + 3) // 4) * 4

    def col_mean(self, X):
        return X.mean(0)

    def pack(self, spec, X, center, role, pad_rows_to=1):
        n, d = X.shape
        kp = self.kp(d)
        rows = ((n + pad_rows_to - 1) // pad_rows_to) * pad_rows_to
        out = torch.zeros(max(rows, 1), kp, dtype=torch.float64)
        v = (X - (center if center is not None else 0.0)) * (1.0 / spec.lengthscale)
        h = -0.5 * (v * v).sum(1)
        out[:n, :d] = v
        if role == ROLE_A:
            out[:n, kp - 2] = h
            out[:n, kp - 1] = 1.0
        else:
            out[:n, kp - 2] = 1.0
            out[:n, kp - 1] = h
        return out

    @staticmethod
    def _kfun(spec, D):
        if spec.family == "rbf":
            return torch.exp(D.clamp_max(0.0))
        r2 = (-2.0 * D).clamp_min(1e-30)
        r = r2.sqrt()
        if spec.family == "matern52":
            a = math.sqrt(5.0) * r
            return ((a + 1.0) + (5.0 / 3.0) * r2) * torch.exp(-a)
        a = math.sqrt(3.0) * r
        return (a + 1.0) * torch.exp(-a)

    def gram(self, spec, packA, na, packB, nb):
        self._count("gram")
        return spec.outputscale * self._kfun(spec, packA[:na] @ packB[:nb].T)

    def matvec(self, spec, packA, na, packB, nb, v, bias):
        self._count("matvec")
        return bias + spec.outputscale * (self._kfun(spec, packA[:na] @ packB[:nb].T) @ v)

    def regroup_classes(self, T, kept, w_star, tot, out=None):
        self._count("regroup")
        Cn, rows, S = T.shape
        H = S // 2
        kk = kept[:H].to(torch.int64)
        To = torch.empty(Cn // 2, rows, S, dtype=torch.float64)
        for par in (0, 1):
            To[:, :, par * H:(par + 1) * H] = (T[par::2][:, :, kk] * w_star[:H]) / tot[kk]
        if out is not None:
            out.copy_(To)
            return out
        return To

    def project_chunks(self, U, q, m, Xpart, totpart, n_chunks, S, outputscale, out=None, ksplit=None):
        self._count("project_chunks")
        M = torch.cat([totpart[:n_chunks].unsqueeze(1), outputscale * torch.matmul(U, Xpart[:n_chunks])], 1)
        if out is not None:
            out.copy_(M)
            return out
        return M

    def sum_parts(self, parts, out=None):
        acc = parts[0].clone()
        for p in range(1, parts.shape[0]):
            acc = acc + parts[p]
        if out is None:
            return acc
        out.copy_(acc.reshape(out.shape))
        return out

    def tail_weights_geo(self, mu, wx, geo_row, S, out):
        n_full, n_tail, off, Rl = (int(geo_row[k]) for k in (1, 5, 6, 7))
        out.zero_()
        for k in range(n_tail):
            p = n_full + k - off                               # local index of tail point k (held by this rank or not)
            if 0 <= p < Rl:
                out[k] = mu[p] if wx is None else mu[p] * wx[p]
        return out

    def blocksum(self, spec, nys, m, cand, mu, wx, Rl, off, n_full, S, n_chunks, out=None, class_mod=0, class0=0):
        self._count("blocksum")
        Xpart = torch.zeros(n_chunks, m, S, dtype=torch.float64)
        totpart = torch.zeros(n_chunks, S, dtype=torch.float64)
        if Rl == 0:
            if out is not None:
                out[0].copy_(Xpart)
                out[1].copy_(totpart)
                return out
            return Xpart, totpart
        pg = off + torch.arange(Rl)
        sets = torch.where(pg < n_full, pg % S, torch.full_like(pg, S - 1))
        # chunk assignment as documented: contiguous block ranges; tail -> last chunk
        lim = min(off + Rl, n_full)
        if lim > off:
            blk_lo, blk_hi = off // S, (lim + S - 1) // S
        else:
            blk_lo = blk_hi = 0
        per = max(1, -(-(blk_hi - blk_lo) // n_chunks))
        chunk = torch.where(pg < n_full, (pg // S - blk_lo) // per, torch.full_like(pg, n_chunks - 1))
        sel = None
        if class_mod > 0:                                      # residue classes of the GLOBAL block index
            assert off + Rl <= n_full and class0 + n_chunks <= class_mod
            cls = (pg // S) % class_mod - class0
            sel = (cls >= 0) & (cls < n_chunks)                # blocks of other classes are not part of this launch
            chunk = cls.clamp(0, n_chunks - 1)
        w = mu[:Rl] * (wx[:Rl] if wx is not None else 1.0)
        mu_eff = mu[:Rl]
        if sel is not None:
            w = torch.where(sel, w, torch.zeros_like(w))
            mu_eff = torch.where(sel, mu_eff, torch.zeros_like(mu_eff))
        flat = chunk * S + sets
        Xf = torch.zeros(m, n_chunks * S, dtype=torch.float64)
        step = 8192
        for lo in range(0, Rl, step):                          # bounded temporaries
            hi = min(Rl, lo + step)
            Kw = self._kfun(spec, nys[:m] @ cand[lo:hi].T) * w[lo:hi].unsqueeze(0)
            Xf.index_add_(1, flat[lo:hi], Kw)
        Xpart = Xf.reshape(m, n_chunks, S).permute(1, 0, 2).contiguous()
        tf = torch.zeros(n_chunks * S, dtype=torch.float64)
        tf.index_add_(0, flat, mu_eff)
        totpart = tf.reshape(n_chunks, S)
        if out is not None:
            out[0].copy_(Xpart)
            out[1].copy_(totpart)
            return out
        return Xpart, totpart

    def project(self, U, q, m, Xpart, totpart, n_chunks, S, outputscale, ksplit=None):
        self._count("project")
        X = Xpart.sum(0)
        return torch.cat([totpart.sum(0).unsqueeze(0), outputscale * (U @ X)], 0)

    def finalize(self, parts, n_parts, msg_rows, q, S, diagU=None, ld_diag=0, n_diag=0, diag_noise=0.0, diag_wrow=0,
                 diag_tail_row=0, n_tail_diag=0, geo_row=None, tot_out=None):
        self._count("finalize")
        if geo_row is not None:
            n_tail_diag = min(n_tail_diag, int(geo_row[5]))
        msg = parts[0].clone()
        for p in range(1, n_parts):
            msg = msg + parts[p]
        tot = msg[0].clone()
        feat = msg[1:q + 1].clone()
        if diagU is not None:
            wgt = msg[diag_wrow].clone()
            if diag_tail_row:
                tail = msg[diag_tail_row]
                wgt[S - 1] -= tail.sum()                       # the last set's weight over the FULL blocks only
                feat[:, S - 1] += diag_noise * (diagU[:, :n_tail_diag] @ tail[:n_tail_diag])
            if n_diag > 0:
                feat[:, :n_diag] += diag_noise * wgt[:n_diag].unsqueeze(0) * diagU[:, :n_diag]
        XcarT = torch.cat([torch.ones(1, S, dtype=torch.float64), feat / tot.unsqueeze(0)], 0)
        if tot_out is not None:
            tot_out.copy_(tot)
            tot = tot_out
        return XcarT, tot

    def reduction_result(self, M):
        """Same layout as ``HipOps.reduction_result``: one float64 buffer, typed views into it."""
        res = torch.zeros(3 * M + 1, dtype=torch.float64)
        ints = res[2 * M:].view(torch.int32)
        ik = ints[:2 + M]
        return res, dict(w_star=res[:M], tot=res[M:2 * M], info=ik[:2], kept=ik[2:], keep_rank=ints[2 + M:2 + 2 * M])

    def nullspace(self, XcarT, s, M, cluster=True):
        self._count("nullspace")
        return householder_nullspace(XcarT)

    def car_eliminate(self, PhiT, mu, M, s, cluster=True, out=None):
        if out is not None:
            res = self.car_eliminate(PhiT, mu, M, s, cluster)
            for key, val in zip(("keep_rank", "kept", "w_star", "info"), res):
                out[key].copy_(val)
            return out["keep_rank"], out["kept"], out["w_star"], out["info"]
        self._count("car")
        mu = mu.clone()                                        # (read only for the caller, as the C entry)
        Phi = PhiT.T.clone()                                   # [M, M-s]
        status = 0
        for _ in range(M - s):
            col = Phi[:, 0]
            pos = col > 0
            if not bool(pos.any()):
                status = 1
                break
            alpha = torch.zeros(M, dtype=torch.float64)
            alpha[pos] = mu[pos] / col[pos]
            j = torch.arange(M)[pos][torch.argmin(alpha[pos])]
            mu[:] = mu - alpha[j] * col
            mu[j] = 0.0
            Phi = Phi[:, 1:]
            Phi = Phi - torch.matmul(Phi[j].unsqueeze(1), col.unsqueeze(1).T).T / col[j]
            Phi[j, :] = 0.0
        keep = mu > 0
        n_keep = int(keep.sum())
        keep_rank = torch.full((M,), -1, dtype=torch.int32)
        keep_rank[keep] = torch.arange(n_keep, dtype=torch.int32)
        kept = torch.zeros(M, dtype=torch.int32)
        kept[:n_keep] = torch.arange(M, dtype=torch.int32)[keep]
        w_star = torch.zeros(M, dtype=torch.float64)
        w_star[:n_keep] = mu[keep]
        info = torch.tensor([n_keep, status], dtype=torch.int32)
        return keep_rank, kept, w_star, info

    def reweight_compact(self, cand, mu, gid, wx, Rl, off, n_full, S, kp, keep_rank, w_star, tot, n_keep, new_off,
                         new_Rl):
        self._count("compact")
        cand_o = torch.zeros(max(new_Rl, 1), kp, dtype=torch.float64)
        mu_o = torch.zeros(max(new_Rl, 1), dtype=torch.float64)
        gid_o = torch.zeros(max(new_Rl, 1), dtype=torch.int64)
        wx_o = torch.zeros(max(new_Rl, 1), dtype=torch.float64) if wx is not None else None
        if Rl == 0:
            return cand_o, mu_o, gid_o, wx_o
        pg = off + torch.arange(Rl)
        inblk = pg < n_full
        sets = torch.where(inblk, pg % S, torch.full_like(pg, S - 1))
        kr = keep_rank.to(torch.int64)[sets]
        dst = torch.where(inblk, (pg // S) * n_keep + kr, (n_full // S) * n_keep + (pg - n_full)) - new_off
        sel = kr >= 0
        d = dst[sel]
        assert d.numel() == new_Rl and (d.numel() == 0 or (int(d.min()) == 0 and int(d.max()) == new_Rl - 1))
        cand_o[d] = cand[:Rl][sel]
        mu_o[d] = (mu[:Rl][sel] * w_star[kr[sel]]) / tot[sets[sel]]
        gid_o[d] = gid[:Rl][sel]
        if wx is not None:
            wx_o[d] = wx[:Rl][sel]
        return cand_o, mu_o, gid_o, wx_o

    # device-resident round descriptors: on the CPU the table is simply read
    def geo_init(self, n_rounds, R, S, reg_hi, off=0, Rl=None):
        g = torch.zeros(n_rounds, 8, dtype=torch.int64)
        nb = R // S
        g[0] = torch.tensor([R, nb * S, reg_hi, 0, nb, R - nb * S, off, R if Rl is None else Rl], dtype=torch.int64)
        return g

    def round_next(self, geo_row, info, keep_rank, S, class_mode, expect_half, geo_next):
        self._count("round_next")
        from basq_amd._partition import RoundGeometry, next_shard

        R, n_full, off, Rl = int(geo_row[0]), int(geo_row[1]), int(geo_row[6]), int(geo_row[7])
        nb, n_tail = n_full // S, R - n_full
        n_keep, status = int(info[0]), int(info[1])
        viol = int(geo_row[3])
        if status != 0 or (expect_half and 2 * n_keep != S):
            viol = 1
        Rn = 0 if viol else nb * n_keep + (n_tail if int(keep_rank[S - 1]) >= 0 else 0)
        nbn = Rn // S
        reg_blocks = (nbn // class_mode) * class_mode if class_mode > 0 else ((int(geo_row[2]) // S) // 2 if class_mode < 0 else 0)
        kept = [j for j in range(S) if int(keep_rank[j]) >= 0]
        n_off, n_Rl = (0, 0) if viol else next_shard(off, Rl, RoundGeometry.of(R, S), kept)
        geo_next[:] = torch.tensor([Rn, nbn * S, reg_blocks * S, viol, nbn, Rn - nbn * S, n_off, n_Rl], dtype=torch.int64)

    def regroup_round_next(self, T, kept, w_star, tot, out, geo_row, info, keep_rank, S, class_mode, expect_half, geo_next):
        self.regroup_classes(T, kept, w_star, tot, out=out)
        self.round_next(geo_row, info, keep_rank, S, class_mode, expect_half, geo_next)
        return out

    def blocksum_geo(self, spec, nys, m, cand, mu, wx, geo_row, mode, S, n_chunks, out=None, class_mod=0, class0=0):
        R, n_full, reg_hi, off, Rl = (int(geo_row[k]) for k in (0, 1, 2, 6, 7))
        lo, hi = (0, reg_hi) if mode == 1 else ((reg_hi, R) if mode == 2 else ((n_full, R) if mode == 4 else
                                                                  ((reg_hi, n_full) if mode == 5 else (0, R))))
        lo, hi = max(lo, off), min(hi, off + Rl)               # the mode's range, restricted to this rank's shard
        hi = max(hi, lo)
        sk = lo - off
        if mode == 4:                                          # the remainder as a block of its own: point k in set k
            return self.blocksum(spec, nys, m, cand[sk:], mu[sk:], None if wx is None else wx[sk:], hi - lo, lo - n_full, S, S,
                                 n_chunks, out=out)
        return self.blocksum(spec, nys, m, cand[sk:], mu[sk:], None if wx is None else wx[sk:], hi - lo, lo, n_full, S,
                             n_chunks, out=out, class_mod=class_mod, class0=class0)

    def reweight_compact_geo(self, cand, mu, gid, wx, geo_row, geo_next, info, R_max, S, kp, keep_rank, w_star, tot,
                             out_rows, expect_keep=-1):
        n_full, off, Rl = int(geo_row[1]), int(geo_row[6]), int(geo_row[7])
        n_keep = int(info[0])
        if int(geo_row[3]) != 0 or int(info[1]) != 0 or (expect_keep >= 0 and n_keep != expect_keep):
            z = lambda *sh, dt=torch.float64: torch.zeros(*sh, dtype=dt)      # noqa: E731  a violating round writes nothing
            return (z(max(out_rows, 1), kp), z(max(out_rows, 1)), z(max(out_rows, 1), dt=torch.int64),
                    z(max(out_rows, 1)) if wx is not None else None)
        new_off, new_R = int(geo_next[6]), int(geo_next[7])
        c, u, g, w = self.reweight_compact(cand, mu, gid, wx, Rl, off, n_full, S, kp, keep_rank, w_star, tot, n_keep,
                                           new_off, new_R)

        def grow(t):
            if t is None:
                return None
            o = torch.zeros((max(out_rows, 1),) + tuple(t.shape[1:]), dtype=t.dtype)
            o[:new_R] = t[:new_R]
            return o

        return grow(c), grow(u), grow(g), grow(w)

    # -- the irregular candidates of an epoch as message columns (basq_epoch_turn_f64 & co.) --------------------------------
    def side_ops(self):
        return self

    def side_context(self):
        import contextlib

        return contextlib.nullcontext()

    def wait_event(self, ev):
        pass

    def epoch_turn(self, Pin, C, E_in, E_out, kept, keep_rank, w_star, tot, info, geo_row, geo_next, out=None):
        self._count("epoch_turn")
        n, rows, S = Pin.shape
        assert n == C + E_in + 2 and C >= 2
        Pout = torch.zeros(C // 2 + E_out + 2, rows, S, dtype=torch.float64) if out is None else out.zero_()
        self.regroup_classes(Pin[:C].contiguous(), kept, w_star, tot, out=Pout[:C // 2])
        self.round_next(geo_row, info, keep_rank, S, -1, True, geo_next)
        nb, reg_blocks, t = int(geo_row[4]), int(geo_row[2]) // S, int(geo_row[5])
        e, n_keep, kr_last = nb - reg_blocks, int(info[0]), int(keep_rank[S - 1])
        bad = int(geo_row[3]) != 0 or int(info[1]) != 0 or 2 * n_keep != S or not (0 <= e <= E_in)
        if bad:
            return Pout
        Iin, fold, Iout = Pin[C + 1:], Pout[C // 2], Pout[C // 2 + 1:]
        cols = []                                               # next round's columns, in position order
        kk = kept[:n_keep].to(torch.int64)
        for b in range(e):
            cols.append((Iin[b][:, kk] * w_star[:n_keep].unsqueeze(0)) / tot[kk].unsqueeze(0))
        if kr_last >= 0 and t > 0:
            cols.append((Iin[E_in][:, :t] * w_star[kr_last]) / tot[S - 1])
        allc = torch.cat(cols, 1) if cols else torch.zeros(rows, 0, dtype=torch.float64)
        n_irr = allc.shape[1]
        e_n, t_n = n_irr // S, n_irr % S
        if e_n > E_out:
            return Pout
        for b in range(e_n):                                    # the blocks in index order ...
            Iout[b] = allc[:, b * S:(b + 1) * S]
            fold += Iout[b]
        Iout[E_out][:, :t_n] = allc[:, e_n * S:]
        # ... then the tail's columns into set S - 1, in the kernel's order: lane l of a wave adds columns l, l + 64, ..., the 64
        # partial sums meet in an xor butterfly (offsets 32 .. 1), the total is added behind the blocks' sum
        part = torch.zeros(rows, 64, dtype=torch.float64)
        for k in range(t_n):
            part[:, k % 64] += allc[:, e_n * S + k]
        lanes = torch.arange(64)
        for o in (32, 16, 8, 4, 2, 1):
            part = part + part[:, lanes ^ o]
        fold[:, S - 1] += part[:, 0]
        return Pout

    def reweight_compact_rounds(self, cand, mu, gid, wx, geo_rows, outcomes, R_max, S, kp, out_rows, expect_keep=-1):
        self._count("compact_rounds")
        for r, o in enumerate(outcomes):
            g, gn = geo_rows[r], geo_rows[r + 1]
            cand, mu, gid, wx = self.reweight_compact_geo(cand, mu, gid, wx, g, gn, o["info"], R_max, S, kp, o["keep_rank"],
                                                          o["w_star"], o["tot"], out_rows if r == len(outcomes) - 1 else
                                                          max(int(gn[7]), 1), expect_keep)
        return cand, mu, gid, wx

    @staticmethod
    def info_kept_buffer(info, kept):
        return torch.cat([info, kept])

    def init_state(self, Rl, gid0, n_total):
        mu = torch.full((max(Rl, 1),), 1.0 / n_total, dtype=torch.float64)
        gid = gid0 + torch.arange(max(Rl, 1), dtype=torch.int64)
        return mu, gid

    def dense_blocksum(self, Cmat, mu_chunk, pg0, n_full, S, scale, E, square=False, tot=None):
        self._count("dense_sq" if square else "dense")
        m, nc = Cmat.shape
        pg = pg0 + torch.arange(nc)
        sets = torch.where(pg < n_full, pg % S, torch.full_like(pg, S - 1))
        E.index_add_(1, sets, scale * ((Cmat * Cmat) if square else Cmat) * mu_chunk[:nc].unsqueeze(0))
        if tot is not None:
            tot.reshape(-1).index_add_(0, sets, mu_chunk[:nc])

    def gram_into(self, spec, packA, na, packB, nb, out):
        out[:na, :nb] = self.gram(spec, packA, na, packB, nb)
        return out

    def blocksum_sq(self, spec, nys, m, cand, mu, Rl, off, n_full, S, n_chunks, bmatT, kobs, n_obs, noise, class_mod=0,
                    class0=0, out=None):
        """E[j, s] = sum_p (mu_p / 2) cov(j, p)^2, cov = s2 k - B ko (+ noise on entry [kappa][kappa] of each block).
        ``class_mod > 0``: chunk c = the blocks b with b % class_mod == class0 + c; otherwise everything lands in chunk 0."""
        self._count("blocksum_sq")
        Epart = torch.zeros(n_chunks, m, S, dtype=torch.float64) if out is None else out.zero_()
        if Rl == 0:
            return Epart
        step = max(S, (1 << 22) // max(m, 1))
        for p0 in range(0, Rl, step):
            nc = min(step, Rl - p0)
            cov = self.gram(spec, nys, m, cand[p0:p0 + nc], nc) - bmatT[:n_obs, :m].T @ kobs[:n_obs, p0:p0 + nc]
            pg = off + p0 + torch.arange(nc)
            if noise != 0.0:
                kappa = torch.where(pg < n_full, pg % S, pg - n_full)
                hit = kappa < m
                cov[kappa[hit], torch.arange(nc)[hit]] += noise
            sets = torch.where(pg < n_full, pg % S, torch.full_like(pg, S - 1))
            vals = 0.5 * (cov * cov) * mu[p0:p0 + nc].unsqueeze(0)
            if class_mod > 0:
                cls = (pg // S) % class_mod - class0
                for c in range(n_chunks):
                    sel = cls == c
                    Epart[c].index_add_(1, sets[sel], vals[:, sel])
            else:
                Epart[0].index_add_(1, sets, vals)
        return Epart

    def cov_diag(self, spec, nys, m, cand, Rl, off, n_full, S, bmatT, kobs, n_obs, noise):
        """out[p] = noise * cov(nys_kappa(p), x_p) + noise^2 / 2 (0 where kappa >= m): see basq_cov_diag_f64."""
        self._count("cov_diag")
        out = torch.zeros(max(Rl, 1), dtype=torch.float64)
        if Rl == 0:
            return out
        pg = off + torch.arange(Rl)
        kappa = torch.where(pg < n_full, pg % S, pg - n_full)
        hit = kappa < m
        kp = nys.shape[1]
        D = (nys[kappa[hit], :kp] * cand[:Rl][hit, :kp]).sum(1)
        kval = spec.outputscale * self._kfun(spec, D)
        corr = (bmatT[:n_obs][:, kappa[hit]] * kobs[:n_obs, :Rl][:, hit]).sum(0)
        out[:Rl][hit] = noise * (kval - corr) + 0.5 * noise * noise
        return out

    # -- WSABI-M in the descriptor-driven rounds: the same arithmetic with the range read from the descriptor ----------------
    def blocksum_sq_geo(self, spec, nys, m, cand, mu, geo_row, mode, S, n_chunks, bmatT, kobs, n_obs, noise, class_mod=0,
                        class0=0, out=None):
        R, n_full, reg_hi, off, Rl = (int(geo_row[k]) for k in (0, 1, 2, 6, 7))
        lo, hi = (0, reg_hi) if mode == 1 else ((reg_hi, R) if mode == 2 else ((n_full, R) if mode == 4 else (0, R)))
        lo, hi = max(lo, off), min(hi, off + Rl)
        hi = max(hi, lo)
        sk = lo - off
        if mode == 4:
            return self.blocksum_sq(spec, nys, m, cand[sk:], mu[sk:], hi - lo, lo - n_full, S, S, n_chunks, bmatT, kobs[:, sk:],
                                    n_obs, noise, out=out)
        return self.blocksum_sq(spec, nys, m, cand[sk:], mu[sk:], hi - lo, lo, n_full, S, n_chunks, bmatT, kobs[:, sk:], n_obs,
                                noise, class_mod=class_mod, class0=class0, out=out)

    def cov_diag_geo(self, spec, nys, m, cand, geo_row, R_max, S, bmatT, kobs, n_obs, noise):
        n_full, off, Rl = int(geo_row[1]), int(geo_row[6]), int(geo_row[7])
        out = torch.zeros(max(R_max, 1), dtype=torch.float64)
        out[:max(Rl, 1)] = self.cov_diag(spec, nys, m, cand, Rl, off, n_full, S, bmatT, kobs, n_obs, noise)[:max(Rl, 1)]
        return out

    def sq_noise_part_geo(self, mu, val, geo_row, U, q, m, S, rows, sober, out):
        """The noise cross terms as a message part (``basq_sq_noise_part_geo_f64``; the host-geometry form of the same sums is
        ``FusedSums.wsabim_class_round``)."""
        n_full, off, Rl = int(geo_row[1]), int(geo_row[6]), int(geo_row[7])
        out.zero_()
        t0l = min(max(n_full - off, 0), Rl)
        if t0l > 0:
            wv = mu[:t0l] * val[:t0l]
            sets = (off + torch.arange(t0l)) % S
            dvec = torch.zeros(S, dtype=torch.float64).index_add_(0, sets, wv)
            nd = min(m, S)
            out[1:q + 1, :nd] = U[:, :nd] * dvec[:nd]
        if Rl > t0l:
            k0 = off + t0l - n_full
            k1 = min(k0 + (Rl - t0l), m)
            if k1 > k0:
                dt = mu[t0l:t0l + (k1 - k0)] * val[t0l:t0l + (k1 - k0)]
                out[1:q + 1, S - 1] += U[:, k0:k1] @ dt
                if sober:
                    out[1:q + 1, k0:k1] += U[:, k0:k1] * dt
        return out

    def box_muller(self, u, u_tail=None):
        def bm(v):
            blk = v.view(-1, 16)
            r = torch.sqrt(-2 * torch.log(1 - blk[:, :8]))
            th = 2.0 * math.pi * blk[:, 8:]
            return torch.cat([r * torch.cos(th), r * torch.sin(th)], 1).reshape(-1)

        n = u.shape[0]
        out = torch.empty(n, dtype=torch.float64)
        out[: n // 16 * 16] = bm(u[: n // 16 * 16])
        if u_tail is not None:
            out[n - 16:] = bm(u_tail)
        return out

    def chol_inv(self, G, rel_tol=1e-12):
        self._count("chol_inv")
        q = G.shape[0]
        L, inf = torch.linalg.cholesky_ex(G)
        bad = int(inf.item())
        d = torch.diagonal(L)
        if bad == 0 and bool((d * d <= rel_tol * torch.diagonal(G).max()).any()):
            bad = int(torch.nonzero(d * d <= rel_tol * torch.diagonal(G).max())[0]) + 1
        info = torch.tensor([bad], dtype=torch.int32)
        if bad:
            return torch.zeros(q, q, dtype=torch.float64), info
        W = torch.linalg.solve_triangular(L, torch.eye(q, dtype=torch.float64), upper=False).T.contiguous()
        G.copy_(torch.tril(L) + torch.triu(G, 1))
        return W, info

    CHOL_FACTOR_MAX_Q = 200
    TRSM_MAX_Q = 318

    def chol_factor(self, G, rel_tol=1e-12):
        self._count("chol_factor")
        L, inf = torch.linalg.cholesky_ex(G)
        bad = int(inf.item())
        d = torch.diagonal(L)
        if bad == 0 and bool((d * d <= rel_tol * torch.diagonal(G).max()).any()):
            bad = int(torch.nonzero(d * d <= rel_tol * torch.diagonal(G).max())[0]) + 1
        if not bad:
            G.copy_(torch.tril(L) + torch.triu(G, 1))
        return torch.tensor([bad], dtype=torch.int32)

    def trsm_rows(self, X, L):
        self._count("trsm_rows")
        return torch.linalg.solve_triangular(torch.tril(L), X.T, upper=False).T.contiguous()

    def matmul(self, A, B):
        return torch.matmul(A, B)

    def gemm(self, A, B, alpha=1.0):
        return alpha * (A @ B)

    def to_host(self, t, tag="d2h"):
        return t

    def to_host_async(self, t, tag="d2h"):
        return t, _DoneEvent()

    def from_host(self, t, tag="h2d"):
        return t.contiguous()

    def host_uniform(self, n, tag):
        return torch.rand(n, dtype=torch.float64)

    def from_pinned(self, buf):
        return buf

    def synchronize(self):
        pass

    def record_event(self, timing=True):
        return None
