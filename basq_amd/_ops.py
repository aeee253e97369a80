"""Device operations of the recombination engine: thin, checked wrappers over the C ABI.

Every method takes/returns torch tensors that live on one HIP device; torch is used only
for memory, streams and a few small dense products outside the per-batch loop (posterior set-up).
All pairwise-kernel work, the range finder's tall-skinny GEMMs, the Nystrom contraction,
the elimination and the compaction run in ``libbasq_hip.so``.

The engine (``_batch.py`` / ``_engine.py``) is written against this interface so that the CPU tests can
drive its host logic (sharding, offsets, collectives) with a stand-in defined under
``tests/`` -- the product never does.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import KernelSpecC, ROLE_A, ROLE_B, check


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


_COPY_STREAMS = {}               # device index -> the copy stream of ``HipOps.from_pinned_side``
_SIDE_OPS = {}                   # device index -> HipOps on that device's side compute stream (``HipOps.side_ops``)
PROJECT_KSPLIT_MAX = 64          # K slices of the projection GEMM (A/B: tools/ab_engine.py --module _ops)


class HipOps:
    """Operations on ``device`` (a ``torch.device('cuda', i)``)."""

    name = "hip"

    def __init__(self, device, stream=None):
        """``stream``: a ``torch.cuda.Stream`` every launch, copy and wait of this object goes to (one per batch in flight:
        ``RecombinationEngine.run_many``); None = torch's current stream at the time of each call.  The caches of an
        instance (pinned staging buffers, the cluster kernels' flag/ring workspace, the transposed basis) belong to that
        stream: two batches in flight never share one ``HipOps``."""
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.BasqHipError("basq_amd runs on an AMD GPU (torch device type 'cuda'); no CPU path exists")
        if not torch.cuda.is_available():
            raise _lib.BasqHipError("no HIP device visible to torch")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.lib = _lib.load()
        self.stream = stream
        self._bound = None if stream is None else C.c_void_p(stream.cuda_stream)

    # -- helpers -----------------------------------------------------------------------------
    def torch_stream(self):
        return self.stream if self.stream is not None else torch.cuda.current_stream(self.device)

    def _stream(self):
        if self._bound is not None:
            return self._bound
        s = self.__dict__.get("_pinned_stream")
        return s if s is not None else C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def pin_stream(self):
        """Look the current stream up ONCE for a whole batch (the engine launches ~250 kernels per batch and the lookup
        costs more than the launch); ``unpin_stream`` restores the per-call lookup."""
        self._pinned_stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def unpin_stream(self):
        self._pinned_stream = None

    def _chk(self, t, dtype=torch.float64):
        if t.device != self.device and not (t.device.type == "cuda" and t.device.index == self.device.index):
            raise ValueError(f"tensor on {t.device}, expected {self.device}")
        if t.dtype != dtype or not t.is_contiguous():
            raise ValueError("expected a contiguous %s tensor" % dtype)
        return t

    def empty(self, *shape, dtype=torch.float64):
        return torch.empty(*shape, dtype=dtype, device=self.device)

    def zeros(self, *shape, dtype=torch.float64):
        return torch.zeros(*shape, dtype=dtype, device=self.device)

    def to_device(self, t, dtype=None):
        return t.to(device=self.device, dtype=dtype or t.dtype).contiguous()

    @staticmethod
    def spec_c(spec) -> KernelSpecC:
        return KernelSpecC(_lib.FAMILY[spec.family], int(spec.d), float(spec.lengthscale), float(spec.outputscale),
                           _lib.SPEC_ACCURATE_EXP if getattr(spec, "accurate_exp", False) else 0, 0)

    def kp(self, d: int) -> int:
        kp = self.lib.basq_kp(int(d))
        check(min(kp, 0), "basq_kp")
        return kp

    def shader_clock_mhz(self, n=1, period_us=20):
        """-> [n] device tensor: the shader clock over each of the next n periods on THIS stream (measurement aid: bench.py
        runs it on a second stream beside the block sums)."""
        out = self.empty(n)
        check(self.lib.basq_shader_clock_mhz(_ptr(out), int(n), int(period_us), self._stream()), "basq_shader_clock_mhz")
        return out

    # -- kernels -----------------------------------------------------------------------------
    def col_mean(self, X):
        self._chk(X)
        n, d = X.shape
        mean = self.empty(d)
        check(self.lib.basq_col_mean_f64(_ptr(X), n, d, _ptr(mean), self._stream()), "basq_col_mean_f64")
        return mean

    def pack(self, spec, X, center, role, pad_rows_to: int = 1):
        """-> [rows, kp] with rows = n rounded up to ``pad_rows_to`` (extra rows zero)."""
        self._chk(X)
        n, d = X.shape
        assert d == spec.d
        kp = self.kp(d)
        rows = ((n + pad_rows_to - 1) // pad_rows_to) * pad_rows_to
        out = self.zeros(max(rows, 1), kp) if rows != n else self.empty(max(n, 1), kp)
        sc = self.spec_c(spec)
        check(self.lib.basq_pack_points_f64(C.byref(sc), _ptr(X), n, _ptr(center), role, _ptr(out), self._stream()),
              "basq_pack_points_f64")
        return out

    def gram(self, spec, packA, na, packB, nb):
        K = self.empty(na, nb)
        sc = self.spec_c(spec)
        check(self.lib.basq_gram_f64(C.byref(sc), _ptr(packA), na, _ptr(packB), nb, _ptr(K), nb, self._stream()),
              "basq_gram_f64")
        return K

    def matvec(self, spec, packA, na, packB, nb, v, bias: float):
        """out[i] = bias + sum_j k(A_i, B_j) v_j.  packA must have rows padded to a multiple of 64."""
        self._chk(v)
        out = self.empty(na)
        sc = self.spec_c(spec)
        check(self.lib.basq_kernel_matvec_f64(C.byref(sc), _ptr(packA), na, _ptr(packB), nb, _ptr(v), float(bias),
                                              _ptr(out), self._stream()), "basq_kernel_matvec_f64")
        return out

    def blocksum(self, spec, nys, m, cand, mu, wx, Rl, off, n_full, S, n_chunks, out=None, class_mod=0, class0=0):
        """``out = (Xpart [n_chunks, m, S], totpart [n_chunks, S])``: write into caller-provided (contiguous) slices.

        ``class_mod > 0``: chunk c = the blocks ``b % class_mod == class0 + c`` (residue classes) instead of contiguous
        block ranges; the range must then hold full blocks only."""
        if out is None:
            Xpart = self.empty(n_chunks, m, S)
            totpart = self.empty(n_chunks, S)
        else:
            Xpart, totpart = out
            assert Xpart.is_contiguous() and totpart.is_contiguous()
            assert tuple(Xpart.shape) == (n_chunks, m, S) and tuple(totpart.shape) == (n_chunks, S)
        if Rl == 0:
            Xpart.zero_()
            totpart.zero_()
            return Xpart, totpart
        sc = self.spec_c(spec)
        check(self.lib.basq_blocksum_f64(C.byref(sc), _ptr(nys), m, _ptr(cand), _ptr(mu), _ptr(wx), Rl, off, n_full, S,
                                         n_chunks, int(class_mod), int(class0), _ptr(Xpart), _ptr(totpart),
                                         self._stream()), "basq_blocksum_f64")
        return Xpart, totpart

    def regroup_classes(self, T, kept, w_star, tot, out=None):
        """Class partials (or class messages) of the next round from this round's (``basq_regroup_classes_f64``):
        ``T [C, rows, S]`` -> ``[C/2, rows, S]`` (``out``: a caller-provided contiguous slice).  ``kept`` int32 (first
        S/2 entries), ``w_star``, ``tot`` on the device."""
        Cn, rows, S = T.shape
        assert T.is_contiguous()
        To = out if out is not None else self.empty(Cn // 2, rows, S)
        assert To.is_contiguous() and tuple(To.shape) == (Cn // 2, rows, S)
        check(self.lib.basq_regroup_classes_f64(_ptr(T), rows, S, Cn, _ptr(kept), _ptr(w_star), _ptr(tot), _ptr(To),
                                                self._stream()), "basq_regroup_classes_f64")
        return To

    def project_chunks(self, U, q, m, Xpart, totpart, n_chunks, S, outputscale, out=None, ksplit=None):
        """Per-chunk messages ``[n_chunks, q + 1, S]``: row 0 = ``totpart[c]``, rows 1.. = ``outputscale * U @ Xpart[c]``."""
        self._chk(U)
        if ksplit is None:
            # (128 slices for the single-chunk launches were tried: GEMM 67 -> 63 us, but the slab reduction 16 -> 31 us)
            ksplit = max(1, min(PROJECT_KSPLIT_MAX, m // 128))
        work = self.empty(n_chunks * ksplit * q * S)
        if out is None:
            out = self.empty(n_chunks, q + 1, S)
        assert out.is_contiguous() and tuple(out.shape) == (n_chunks, q + 1, S)
        check(self.lib.basq_project_chunks_f64(_ptr(self._transposed(U)), q, m, _ptr(Xpart), _ptr(totpart), n_chunks, S, float(outputscale),
                                               ksplit, _ptr(work), _ptr(out), self._stream()), "basq_project_chunks_f64")
        return out

    def sum_parts(self, parts, out=None):
        """``parts [n, ...] -> parts.sum(0)`` in index order (``out``: a contiguous destination of that size)."""
        assert parts.is_contiguous()
        if out is None:
            out = self.empty(*parts.shape[1:])
        assert out.is_contiguous() and out.numel() == parts[0].numel()
        check(self.lib.basq_sum_parts_f64(_ptr(parts), parts.shape[0], out.numel(), _ptr(out), self._stream()),
              "basq_sum_parts_f64")
        return out

    def _transposed(self, U):
        """``U.t().contiguous()``, cached per tensor: the contraction kernels read the basis transposed (``[m, q]``)."""
        key = (U.data_ptr(), U._version, tuple(U.shape))
        hit = self.__dict__.get("_ut_cache")
        if hit is None or hit[0] != key:
            hit = self.__dict__["_ut_cache"] = (key, U.t().contiguous(), U)      # keeps U alive: the pointer stays unique
        return hit[1]

    def project(self, U, q, m, Xpart, totpart, n_chunks, S, outputscale, ksplit=None):
        self._chk(U)
        if ksplit is None:
            ksplit = max(1, min(PROJECT_KSPLIT_MAX, m // 128))
        work = self.empty((m * S if n_chunks > 1 else 0) + ksplit * q * S)
        out = self.empty(q + 1, S)
        check(self.lib.basq_project_f64(_ptr(self._transposed(U)), q, m, _ptr(Xpart), _ptr(totpart), n_chunks, S, float(outputscale),
                                        ksplit, _ptr(work), _ptr(out), self._stream()), "basq_project_f64")
        return out

    def finalize(self, parts, n_parts, msg_rows, q, S, diagU=None, ld_diag=0, n_diag=0, diag_noise=0.0, diag_wrow=0,
                 diag_tail_row=0, n_tail_diag=0, geo_row=None, tot_out=None):
        """``geo_row``: descriptor-driven round -- the tail length comes from the device, ``n_tail_diag`` is its cap.
        ``tot_out``: where the set weights go (a contiguous ``[S]`` slice of a ``reduction_result`` buffer)."""
        self._chk(parts)
        XcarT = self.empty(q + 1, S)
        tot = self.empty(S) if tot_out is None else tot_out
        assert tot.is_contiguous() and tot.numel() == S
        check(self.lib.basq_finalize_geo_f64(_ptr(parts), n_parts, msg_rows, q, S, _ptr(diagU), ld_diag, n_diag,
                                             float(diag_noise), diag_wrow, diag_tail_row, n_tail_diag, _ptr(geo_row),
                                             _ptr(XcarT), _ptr(tot), self._stream()),
              "basq_finalize_geo_f64")
        return XcarT, tot

    def tail_weights_geo(self, mu, wx, geo_row, S, out):
        """``out[k] = mu * wx`` of tail point k of a descriptor-driven round (zero beyond the tail); ``out`` = a row of S."""
        assert out.is_contiguous() and out.numel() == S
        check(self.lib.basq_tail_weights_geo_f64(_ptr(mu), _ptr(wx), _ptr(geo_row), S, _ptr(out), self._stream()),
              "basq_tail_weights_geo_f64")
        return out

    def nullspace(self, XcarT, s, M, cluster=True):
        """Rows s..M-1 of the full ``Vh`` of ``svd(XcarT [s, M])`` (``_rchq.py:140-143``) -> PhiT [M-s, M].

        Shapes that need the 8-work-group cluster kernels (M > 256) attach ``PhiT.ns_info`` (device int32[1]: 2 = a
        cluster spin timed out), which ``car_eliminate`` folds into its status word; ``cluster=False`` selects the
        single-work-group kernels (the retry path after such a time-out)."""
        self._chk(XcarT)
        V = self.empty(s, M)
        tau = self.empty(s)
        PhiT = self.empty(M - s, M)
        ws = self._reduction_ws(s, M) if cluster else None
        info = self.empty(1, dtype=torch.int32) if ws is not None else None
        check(self.lib.basq_nullspace_f64(_ptr(XcarT), s, M, _ptr(V), _ptr(tau), _ptr(PhiT), _ptr(ws), _ptr(info),
                                          self._stream()), "basq_nullspace_f64")
        if info is not None:
            PhiT.ns_info = info
        return PhiT

    def _reduction_ws(self, s, M):
        """Cached workspace of the cluster kernels (message ring + flag words), None when the shape needs none."""
        n = int(self.lib.basq_reduction_ws_doubles(int(s), int(M)))
        if n <= 0:
            return None
        cache = self.__dict__.setdefault("_ws_cache", {})
        buf = cache.get(n)
        if buf is None:
            buf = cache[n] = self.zeros(n)
        return buf

    def reduction_result(self, M):
        """ONE contiguous buffer for the outcome of a round's reduction + typed views into it -- what the rank that ran the
        reduction broadcasts to the others (owner-rank mode, ``_batch.py``): ``res [3 M + 1]`` float64 =
        ``w_star [M] | tot [M] | int32: info [2], kept [M], keep_rank [M]``.  -> ``(res, views)``."""
        res = self.empty(3 * M + 1)
        ints = res[2 * M:].view(torch.int32)                    # 2 M + 2 int32 words
        ik = ints[:2 + M]
        info, kept = ik[:2], ik[2:]
        info.ik_buffer = ik
        return res, dict(w_star=res[:M], tot=res[M:2 * M], info=info, kept=kept, keep_rank=ints[2 + M:2 + 2 * M])

    def car_eliminate(self, PhiT, mu, M, s, cluster=True, out=None):
        """Destroys PhiT; ``mu`` is read only.  -> (keep_rank[M] i32, kept[s] i32, w_star[s] f64, info[2] i32 = [n_keep, status]);
        status 0 ok, 1 = a null vector without a positive entry, 2 = a cluster kernel (this one or the null space's) timed
        out waiting for its sibling work-groups.  ``out``: the views of ``reduction_result(M)`` to write into."""
        self._chk(PhiT)
        self._chk(mu)
        if M > self.NULLSPACE_MAX_M:
            res = self._car_eliminate_wide(PhiT, mu, M, s)
            if out is None:
                return res
            for key, val in zip(("keep_rank", "kept", "w_star", "info"), res):
                out[key].copy_(val[:out[key].numel()])
            return out["keep_rank"], out["kept"], out["w_star"], out["info"]
        if out is not None:
            keep_rank, kept, w_star, info = out["keep_rank"], out["kept"], out["w_star"], out["info"]
        else:
            keep_rank = self.empty(M, dtype=torch.int32)
            ik = self.empty(2 + M, dtype=torch.int32)          # [info(2) | kept(<=M)]  (<= s unless the elimination
            info, kept = ik[:2], ik[2:]                         #  stopped early, status 1)
            info.ik_buffer = ik                                 # (info_kept_buffer: one D2H for both)
            w_star = self.empty(M)
        check(self.lib.basq_car_eliminate_f64(_ptr(PhiT), _ptr(mu), M, s, _ptr(keep_rank), _ptr(kept), _ptr(w_star),
                                              _ptr(info), _ptr(self._reduction_ws(s, M) if cluster else None),
                                              self._stream()), "basq_car_eliminate_f64")
        ns_info = getattr(PhiT, "ns_info", None)
        if ns_info is not None:                                 # a timed-out null space must not pass as status 1
            torch.maximum(info[1:2], ns_info, out=info[1:2])
        return keep_rank, kept, w_star, info

    def _car_eliminate_wide(self, PhiT, mu, M, s):
        """``basq_car_eliminate_f64``'s contract for M = 2 * num_pts > 1024, beyond the kernels' one-thread-per-set layout:
        the steps of ``BASQ/_rchq.py:146-171`` as device tensor operations in the reference's own order (one small launch
        per operation, ~10 per step: the envelope path, an order of magnitude slower than the kernel)."""
        Phi = PhiT.t().contiguous()                             # [M, M - s], columns = null vectors
        status = 0
        for _ in range(M - s):
            col = Phi[:, 0]
            pos = col > 0
            if not bool(pos.any()):                             # :152 would raise (argmin of an empty tensor)
                status = 1
                break
            alpha = torch.where(pos, mu / col, torch.full_like(mu, float("inf")))
            j = torch.argmin(alpha)
            mu = mu - alpha[j] * col                            # :156
            mu[j] = 0.0
            cj = col[j]
            Phi = Phi[:, 1:] - torch.outer(col, Phi[j, 1:]) / cj      # :165-167 (outer product, then the division)
            Phi[j, :] = 0.0
        keep = mu > 0
        n_keep = int(keep.sum())
        idx = torch.nonzero(keep).reshape(-1).to(torch.int32)
        keep_rank = torch.full((M,), -1, dtype=torch.int32, device=self.device)
        keep_rank[idx.long()] = torch.arange(n_keep, dtype=torch.int32, device=self.device)
        ik = torch.zeros(2 + M, dtype=torch.int32, device=self.device)
        ik[0], ik[1] = n_keep, status
        ik[2:2 + n_keep] = idx
        w_star = self.zeros(M)
        w_star[:n_keep] = mu[keep]
        info, kept = ik[:2], ik[2:]
        info.ik_buffer = ik
        return keep_rank, kept, w_star, info

    @staticmethod
    def info_kept_buffer(info, kept):
        """``[info(2) | kept]`` as ONE tensor (``car_eliminate`` hands out views of such a buffer)."""
        ik = getattr(info, "ik_buffer", None)
        return ik if ik is not None else torch.cat([info, kept])

    NULLSPACE_MAX_M = 1024           # largest 2 * num_pts of basq_nullspace_f64 / basq_car_eliminate_f64

    def reweight_compact(self, cand, mu, gid, wx, Rl, off, n_full, S, kp, keep_rank, w_star, tot, n_keep, new_off,
                         new_Rl):
        cand_o = self.empty(max(new_Rl, 1), kp)
        mu_o = self.empty(max(new_Rl, 1))
        gid_o = self.empty(max(new_Rl, 1), dtype=torch.int64)
        wx_o = self.empty(max(new_Rl, 1)) if wx is not None else None
        check(self.lib.basq_reweight_compact_f64(_ptr(cand), _ptr(mu), _ptr(gid), _ptr(wx), Rl, off, n_full, S, kp,
                                                 _ptr(keep_rank), _ptr(w_star), _ptr(tot), n_keep, new_off,
                                                 _ptr(cand_o), _ptr(mu_o), _ptr(gid_o), _ptr(wx_o), self._stream()),
              "basq_reweight_compact_f64")
        return cand_o, mu_o, gid_o, wx_o

    # -- device-resident round descriptors (basq_round_next_i64 and the *_geo entries) -----------------------------
    def geo_init(self, n_rounds, R, S, reg_hi, off=0, Rl=None):
        """Descriptor table ``[n_rounds, 8]`` (int64, device) with row 0 = the first round's geometry and this rank's shard
        ``[off, off + Rl)`` of the R live positions (default: all of them)."""
        nb = R // S
        host = self._pinned((n_rounds, 8), torch.int64, "geo")
        host.zero_()
        host[0] = torch.tensor([R, nb * S, reg_hi, 0, nb, R - nb * S, off, R if Rl is None else Rl], dtype=torch.int64)
        return host.to(self.device, non_blocking=True)

    def round_next(self, geo_row, info, keep_rank, S, class_mode, expect_half, geo_next):
        check(self.lib.basq_round_next_i64(_ptr(geo_row), _ptr(info), _ptr(keep_rank), S, int(class_mode),
                                           1 if expect_half else 0, _ptr(geo_next), self._stream()), "basq_round_next_i64")

    def regroup_round_next(self, T, kept, w_star, tot, out, geo_row, info, keep_rank, S, class_mode, expect_half, geo_next):
        """``regroup_classes`` + ``round_next`` in one launch (``basq_regroup_round_next_f64``)."""
        Cn, rows, S_ = T.shape
        assert T.is_contiguous() and out.is_contiguous() and tuple(out.shape) == (Cn // 2, rows, S_) and S_ == S
        check(self.lib.basq_regroup_round_next_f64(_ptr(T), rows, S, Cn, _ptr(kept), _ptr(w_star), _ptr(tot), _ptr(out),
                                                   _ptr(geo_row), _ptr(info), _ptr(keep_rank), int(class_mode),
                                                   1 if expect_half else 0, _ptr(geo_next), self._stream()),
              "basq_regroup_round_next_f64")
        return out

    def blocksum_geo(self, spec, nys, m, cand, mu, wx, geo_row, mode, S, n_chunks, out=None, class_mod=0, class0=0):
        """``blocksum`` with the candidate range read from a round descriptor (``mode`` 1: regular region, 2: the rest,
        3: everything)."""
        if out is None:
            Xpart, totpart = self.empty(n_chunks, m, S), self.empty(n_chunks, S)
        else:
            Xpart, totpart = out
            assert Xpart.is_contiguous() and totpart.is_contiguous()
            assert tuple(Xpart.shape) == (n_chunks, m, S) and tuple(totpart.shape) == (n_chunks, S)
        sc = self.spec_c(spec)
        check(self.lib.basq_blocksum_geo_f64(C.byref(sc), _ptr(nys), m, _ptr(cand), _ptr(mu), _ptr(wx), _ptr(geo_row),
                                             int(mode), S, n_chunks, int(class_mod), int(class0), _ptr(Xpart),
                                             _ptr(totpart), self._stream()), "basq_blocksum_geo_f64")
        return Xpart, totpart

    def reweight_compact_geo(self, cand, mu, gid, wx, geo_row, geo_next, info, R_max, S, kp, keep_rank, w_star, tot,
                             out_rows, expect_keep=-1):
        """``reweight_compact`` with the shard and the counts read from the descriptors (this round's and the next
        round's); launch sized for ``R_max`` local candidates, outputs sized ``out_rows`` for ``expect_keep`` kept sets (a
        round that violates that writes nothing, see ``basq_reweight_compact_geo_f64``)."""
        cand_o = self.empty(max(out_rows, 1), kp)
        mu_o = self.empty(max(out_rows, 1))
        gid_o = self.empty(max(out_rows, 1), dtype=torch.int64)
        wx_o = self.empty(max(out_rows, 1)) if wx is not None else None
        check(self.lib.basq_reweight_compact_geo_f64(_ptr(cand), _ptr(mu), _ptr(gid), _ptr(wx), _ptr(geo_row),
                                                     _ptr(geo_next), _ptr(info), int(max(R_max, 1)), S, kp, _ptr(keep_rank), _ptr(w_star), _ptr(tot),
                                                     int(max(out_rows, 1)), int(expect_keep), _ptr(cand_o), _ptr(mu_o), _ptr(gid_o), _ptr(wx_o), self._stream()),
              "basq_reweight_compact_geo_f64")
        return cand_o, mu_o, gid_o, wx_o

    # -- epochs without a pairwise evaluation inside: the irregular candidates as message columns (ABI 15) -----------
    def epoch_turn(self, Pin, C, E_in, E_out, kept, keep_rank, w_star, tot, info, geo_row, geo_next, out=None):
        """The launch behind an elimination inside an epoch -> next round's buffer ``[C / 2 | fold | E_out | tail]`` and the next
        round's descriptor (``basq_epoch_turn_f64``)."""
        n, rows, S = Pin.shape
        assert Pin.is_contiguous() and n == C + E_in + 2 and C >= 2
        Pout = self.empty(C // 2 + E_out + 2, rows, S) if out is None else out
        assert Pout.is_contiguous() and tuple(Pout.shape) == (C // 2 + E_out + 2, rows, S)
        check(self.lib.basq_epoch_turn_f64(_ptr(Pin), int(C), int(E_in), _ptr(Pout), int(E_out), rows, S, _ptr(kept), _ptr(keep_rank),
                                           _ptr(w_star), _ptr(tot), _ptr(info), _ptr(geo_row), _ptr(geo_next), self._stream()),
              "basq_epoch_turn_f64")
        return Pout

    def reweight_compact_rounds(self, cand, mu, gid, wx, geo_rows, outcomes, R_max, S, kp, out_rows, expect_keep=-1):
        """The compactions of ``len(outcomes)`` consecutive rounds in one launch (``outcomes``: per round the dict of
        ``car_eliminate`` -- keep_rank, w_star, tot, info; ``geo_rows``: the descriptor table from the first of them on)."""
        n = len(outcomes)
        cand_o = self.empty(max(out_rows, 1), kp)
        mu_o = self.empty(max(out_rows, 1))
        gid_o = self.empty(max(out_rows, 1), dtype=torch.int64)
        wx_o = self.empty(max(out_rows, 1)) if wx is not None else None
        arr = lambda key: (C.c_void_p * n)(*[o[key].data_ptr() for o in outcomes])      # noqa: E731
        check(self.lib.basq_reweight_compact_rounds_f64(_ptr(cand), _ptr(mu), _ptr(gid), _ptr(wx), _ptr(geo_rows), n, arr("keep_rank"),
                                                        arr("w_star"), arr("tot"), arr("info"), int(max(R_max, 1)), S, kp,
                                                        int(max(out_rows, 1)), int(expect_keep), _ptr(cand_o), _ptr(mu_o), _ptr(gid_o),
                                                        _ptr(wx_o), self._stream()), "basq_reweight_compact_rounds_f64")
        return cand_o, mu_o, gid_o, wx_o

    def init_state(self, Rl, gid0, n_total):
        mu = self.empty(max(Rl, 1))
        gid = self.empty(max(Rl, 1), dtype=torch.int64)
        check(self.lib.basq_init_state_f64(_ptr(mu), _ptr(gid), Rl, gid0, n_total, self._stream()),
              "basq_init_state_f64")
        return mu, gid

    def dense_blocksum(self, Cmat, mu_chunk, pg0, n_full, S, scale, E, square=False, tot=None):
        """E[j, s] += scale * sum_{p in chunk, set(p)=s} mu_p * C[j, p]  (``square``: ... * C[j, p]^2); in place on E.
        ``tot`` (optional, ``[S]``): ``tot[s] += sum mu_p`` over the same candidates, in the same launch.

        ``Cmat`` may be a row-strided view (``stride(1) == 1``) of a wider buffer."""
        self._chk(E)
        if Cmat.dtype != torch.float64 or Cmat.stride(1) != 1:
            raise ValueError("expected a float64 matrix with unit column stride")
        if tot is not None:
            self._chk(tot)
            assert tot.numel() == S
        m, nc = Cmat.shape
        check(self.lib.basq_dense_blocksum_f64(_ptr(Cmat), m, nc, Cmat.stride(0), _ptr(mu_chunk), pg0, n_full, S,
                                               float(scale), 1 if square else 0, _ptr(E), _ptr(tot), self._stream()),
              "basq_dense_blocksum_f64")

    def blocksum_sq(self, spec, nys, m, cand, mu, Rl, off, n_full, S, n_chunks, bmatT, kobs, n_obs, noise, class_mod=0,
                    class0=0, out=None):
        """WSABI-M's squared-covariance block sums, fused (``basq_blocksum_sq_f64``) -> ``Epart [n_chunks, m, S]``.

        ``bmatT [n_obs4, >= m padded to 64]``, ``kobs [n_obs4, >= Rl]`` (zero rows beyond ``n_obs``).  ``class_mod > 0``: chunk
        c = the blocks ``b % class_mod == class0 + c`` (full blocks only), as for ``blocksum``."""
        self._chk(bmatT)
        if kobs.dtype != torch.float64 or kobs.stride(1) != 1:  # (a column-offset view of a wider buffer is fine)
            raise ValueError("expected a float64 matrix with unit column stride")
        Epart = self.empty(n_chunks, m, S) if out is None else out
        assert Epart.is_contiguous() and tuple(Epart.shape) == (n_chunks, m, S)
        if Rl == 0:
            return Epart.zero_()
        sc = self.spec_c(spec)
        check(self.lib.basq_blocksum_sq_f64(C.byref(sc), _ptr(nys), m, _ptr(cand), _ptr(mu), Rl, off, n_full, S, n_chunks,
                                            int(class_mod), int(class0), _ptr(bmatT), bmatT.stride(0), _ptr(kobs),
                                            kobs.stride(0), int(n_obs), float(noise), _ptr(Epart), self._stream()),
              "basq_blocksum_sq_f64")
        return Epart

    def cov_diag(self, spec, nys, m, cand, Rl, off, n_full, S, bmatT, kobs, n_obs, noise):
        """``out[p] = noise * cov(nys_kappa(p), x_p) + noise^2 / 2`` for the Rl local candidates (``basq_cov_diag_f64``):
        the noise cross terms of WSABI-M's squared covariance, which sit on one Nystrom row per candidate."""
        out = self.empty(max(Rl, 1))
        if Rl == 0:
            return out.zero_()
        sc = self.spec_c(spec)
        check(self.lib.basq_cov_diag_f64(C.byref(sc), _ptr(nys), m, _ptr(cand), Rl, off, n_full, S, _ptr(bmatT),
                                         bmatT.stride(0), _ptr(kobs), kobs.stride(0), int(n_obs), float(noise), _ptr(out),
                                         self._stream()), "basq_cov_diag_f64")
        return out

    # -- WSABI-M in the descriptor-driven rounds ---------------------------------------------------------------------
    def blocksum_sq_geo(self, spec, nys, m, cand, mu, geo_row, mode, S, n_chunks, bmatT, kobs, n_obs, noise, class_mod=0,
                        class0=0, out=None):
        """``blocksum_sq`` with the candidate range read from a round descriptor (modes as for ``blocksum_geo``); ``kobs`` holds
        this rank's live candidates from its first one."""
        self._chk(bmatT)
        if kobs.dtype != torch.float64 or kobs.stride(1) != 1:
            raise ValueError("expected a float64 matrix with unit column stride")
        Epart = self.empty(n_chunks, m, S) if out is None else out
        assert Epart.is_contiguous() and tuple(Epart.shape) == (n_chunks, m, S)
        sc = self.spec_c(spec)
        check(self.lib.basq_blocksum_sq_geo_f64(C.byref(sc), _ptr(nys), m, _ptr(cand), _ptr(mu), _ptr(geo_row), int(mode), S,
                                                n_chunks, int(class_mod), int(class0), _ptr(bmatT), bmatT.stride(0), _ptr(kobs),
                                                kobs.stride(0), int(n_obs), float(noise), _ptr(Epart), self._stream()),
              "basq_blocksum_sq_geo_f64")
        return Epart

    def cov_diag_geo(self, spec, nys, m, cand, geo_row, R_max, S, bmatT, kobs, n_obs, noise):
        """``cov_diag`` for the shard a round descriptor names; launch and output sized for ``R_max`` candidates."""
        out = self.empty(max(R_max, 1))
        sc = self.spec_c(spec)
        check(self.lib.basq_cov_diag_geo_f64(C.byref(sc), _ptr(nys), m, _ptr(cand), _ptr(geo_row), int(max(R_max, 1)), S,
                                             _ptr(bmatT), bmatT.stride(0), _ptr(kobs), kobs.stride(0), int(n_obs), float(noise),
                                             _ptr(out), self._stream()), "basq_cov_diag_geo_f64")
        return out

    def sq_noise_part_geo(self, mu, val, geo_row, U, q, m, S, rows, sober, out):
        """WSABI-M's noise cross terms as a part of the round's message (``basq_sq_noise_part_geo_f64``) -> ``out [rows, S]``."""
        self._chk(U)
        assert out.is_contiguous() and tuple(out.shape) == (rows, S)
        ws = self.empty(int(self.lib.basq_sq_noise_part_ws_doubles(int(S))))
        check(self.lib.basq_sq_noise_part_geo_f64(_ptr(mu), _ptr(val), _ptr(geo_row), _ptr(U), U.stride(0), q, m, S, rows,
                                                  1 if sober else 0, _ptr(ws), _ptr(out), self._stream()),
              "basq_sq_noise_part_geo_f64")
        return out

    def gram_into(self, spec, packA, na, packB, nb, out):
        """``out[:na, :nb] = outputscale * k(A, B)`` for a caller-provided row-major buffer (row stride ``out.stride(0)``)."""
        self._chk(out)
        assert out.stride(1) == 1 and out.shape[0] >= na and out.shape[1] >= nb
        sc = self.spec_c(spec)
        check(self.lib.basq_gram_f64(C.byref(sc), _ptr(packA), na, _ptr(packB), nb, _ptr(out), out.stride(0), self._stream()),
              "basq_gram_f64")
        return out

    def box_muller(self, u, u_tail=None):
        """Normals from torch's uniforms (see basq_box_muller_f64); u on the device, n >= 16."""
        self._chk(u)
        out = self.empty(u.shape[0])
        check(self.lib.basq_box_muller_f64(_ptr(u), u.shape[0], _ptr(u_tail), _ptr(out), self._stream()),
              "basq_box_muller_f64")
        return out

    CHOL_SQUARE_MAX_Q = 142          # q (q|1) + q doubles fit in 160 KB of LDS
    CHOL_PACKED_MAX_Q = 200          # q (q+1)/2 + q doubles fit

    def chol_inv(self, G, rel_tol=1e-12):
        """In place: G -> L (lower).  Returns (W = L^{-T}, info[1] int32 on device)."""
        self._chk(G)
        q = G.shape[0]
        info = self.empty(1, dtype=torch.int32)
        if self.CHOL_SQUARE_MAX_Q < q <= self.CHOL_PACKED_MAX_Q:
            # the square no longer fits in LDS: factor in the packed-triangle kernel, W = L^{-T} by a library
            # triangular solve (plumbing); a failed factorisation (info != 0) leaves garbage in W, as the kernels do
            check(self.lib.basq_chol_inv_f64(_ptr(G), q, None, _ptr(info), float(rel_tol), self._stream()),
                  "basq_chol_inv_f64")
            eye = torch.eye(q, dtype=torch.float64, device=self.device)
            W = torch.linalg.solve_triangular(torch.tril(G).t(), eye, upper=True)
            return torch.nan_to_num(W).contiguous(), info
        W = self.empty(q, q)
        check(self.lib.basq_chol_inv_f64(_ptr(G), q, _ptr(W), _ptr(info), float(rel_tol), self._stream()),
              "basq_chol_inv_f64")
        return W, info

    CHOL_FACTOR_MAX_Q = 200          # packed lower triangle in LDS
    TRSM_MAX_Q = 318

    def chol_factor(self, G, rel_tol=1e-12):
        """In place: G -> L (lower triangle).  Returns info[1] int32 on the device (0 ok, j+1: pivot j too small)."""
        self._chk(G)
        info = self.empty(1, dtype=torch.int32)
        check(self.lib.basq_chol_factor_f64(_ptr(G), G.shape[0], _ptr(info), float(rel_tol), self._stream()),
              "basq_chol_factor_f64")
        return info

    def trsm_rows(self, X, L):
        """``X @ L^-T`` for a tall X [rows, q] (row-major) and the factor left by ``chol_factor``."""
        self._chk(X)
        self._chk(L)
        rows, q = X.shape
        out = self.empty(rows, q)
        check(self.lib.basq_trsm_rows_f64(_ptr(X), q, rows, q, _ptr(L), _ptr(out), q, self._stream()), "basq_trsm_rows_f64")
        return out

    CHOLQR_FUSED_MAX_Q = 200

    def cholqr(self, G, X, rel_tol=1e-12):
        """``chol_factor`` + ``trsm_rows`` in ONE launch (q <= CHOLQR_FUSED_MAX_Q): G -> L in place, returns
        ``(X @ L^-T, info[1])`` -- the solve of a column panel starts as soon as the factor has produced it."""
        self._chk(G)
        self._chk(X)
        rows, q = X.shape
        out = self.empty(rows, q)
        info = self.empty(2, dtype=torch.int32)
        check(self.lib.basq_cholqr_f64(_ptr(G), q, _ptr(info), float(rel_tol), _ptr(X), q, rows, _ptr(out), q, self._stream()),
              "basq_cholqr_f64")
        return out, info[:1]

    def gemm(self, A, B, alpha=1.0):
        """C = alpha * A @ B on the f64 matrix cores (own kernel)."""
        self._chk(A)
        self._chk(B)
        Mr, K = A.shape
        K2, N = B.shape
        assert K == K2
        Cm = self.empty(Mr, N)
        check(self.lib.basq_gemm_f64(_ptr(A), K, _ptr(B), N, _ptr(Cm), N, Mr, N, K, float(alpha), self._stream()),
              "basq_gemm_f64")
        return Cm

    # -- plumbing (small dense products outside the hot loop go through torch) ---------------------
    def matmul(self, A, B):
        return torch.matmul(A, B)

    SKINNY_MAX_N = 208               # widest output of basq_skinny_gemm_f64 (13 column tiles)

    @staticmethod
    def _skinny_ksplit(M, K, N):
        """K slices for ``basq_skinny_gemm_f64``: ~1900 waves per launch -- just under the 2048 wave slots of the chip at
        two waves per SIMD.  6 slices at q = 99 run as fast as 16 (435 / 464 us vs 435 / 462, the two optima of
        profiles/r02_d_skinny_gemm_sweep.txt) and leave 6 instead of 16 slabs to add; slices of at least ten 16-k trips."""
        rows_per_wave = 16 if N > 112 else 32
        row_waves = (M + rows_per_wave - 1) // rows_per_wave
        return max(1, min(round(1900 / row_waves), K // 160))

    def skinny_gemm(self, A, B, trans=False, ksplit=None):
        """``A @ B`` (``trans``: ``A.T @ B``) for a skinny ``B [K, N <= 208]`` on the hand-written f64 MFMA kernel
        (``basq_skinny_gemm_f64``): the range finder's products.  ``A``, ``B`` row-major (row strides free)."""
        if A.dtype != torch.float64 or B.dtype != torch.float64 or A.stride(1) != 1 or B.stride(1) != 1:
            raise ValueError("expected float64 matrices with unit column stride")
        K, M = (A.shape[0], A.shape[1]) if trans else (A.shape[1], A.shape[0])
        N = B.shape[1]
        assert B.shape[0] == K and N <= self.SKINNY_MAX_N
        if ksplit is None:
            ksplit = self._skinny_ksplit(M, K, N)
        out = self.empty(M, N)
        work = self.empty(ksplit * M * N) if ksplit > 1 else None
        check(self.lib.basq_skinny_gemm_f64(_ptr(A), A.stride(0), 1 if trans else 0, M, K, _ptr(B), B.stride(0), N,
                                            int(ksplit), _ptr(work), _ptr(out), self._stream()), "basq_skinny_gemm_f64")
        return out

    # -- host <-> device staging through cached pinned buffers (per-round 160 KB / 80 KB / 0.4 KB copies) ---------
    PIN_CACHE_ENTRIES = 48           # a batch uses ~8 (tag, shape) pairs; a long loop over varying shapes stays bounded

    def _pinned(self, shape, dtype, tag):
        cache = self.__dict__.setdefault("_pin_cache", {})
        key = (tag, tuple(shape), dtype)
        buf = cache.pop(key, None)
        if buf is None:
            buf = torch.empty(tuple(shape), dtype=dtype, pin_memory=True)
            while len(cache) >= self.PIN_CACHE_ENTRIES:         # least recently used first (dicts keep insertion order)
                cache.pop(next(iter(cache)))
        cache[key] = buf                                        # (re)inserted at the end: most recently used
        return buf

    def release_caches(self):
        """Drop the pinned staging buffers, the cluster kernels' workspace and the transposed-basis cache."""
        for k in ("_pin_cache", "_ws_cache", "_ut_cache"):
            self.__dict__.pop(k, None)

    def to_host(self, t, tag="d2h"):
        """Device -> pinned host tensor (synchronises the stream).  The buffer is reused by the next call with
        the same tag and shape: consume it before then."""
        buf = self._pinned(t.shape, t.dtype, tag)
        buf.copy_(t, non_blocking=True)
        # (polling an event instead of this blocking wait was tried: no measurable difference, 41.5 vs 41.6 batches/s)
        self.torch_stream().synchronize()
        return buf

    def to_host_async(self, t, tag="d2h"):
        """Device -> pinned host tensor, enqueued on the current stream WITHOUT waiting: returns ``(buf, event)``;
        ``event.synchronize()`` makes ``buf`` valid.  Work enqueued afterwards overlaps with the host's use of it."""
        buf = self._pinned(t.shape, t.dtype, tag)
        buf.copy_(t, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(self.torch_stream())
        return buf, ev

    def host_uniform(self, n, tag):
        """``torch.rand(n, float64)`` from the CPU global generator, written straight into a cached pinned buffer."""
        buf = self._pinned((n,), torch.float64, tag)
        return torch.rand(n, dtype=torch.float64, out=buf)

    def from_pinned(self, buf):
        """Pinned host tensor -> device (asynchronous on the current stream)."""
        return buf.to(self.device, non_blocking=True)

    def from_pinned_side(self, buf):
        """The same copy on a stream of its own (the copy engine then runs beside whatever the launch stream is busy with --
        the round-1 block sums, for the range finder's 8 MB of uniforms); the launch stream waits for it on the device."""
        # ONE copy stream per device for the life of the process: the caching allocator keeps a block pool per stream, and
        # ``recombination()`` builds a fresh HipOps per call -- a stream per call left a 20-MB segment behind on each of torch's 32
        # pooled streams before they came round again (tools/stall_probe.py: one hipMalloc per batch for the first 32 batches)
        # -- and batches in flight share it too: a copy stream per slot was measured and LOST (N = 2e4, two in flight: 250 vs 357
        # batches/s on one box; HIP maps streams onto a few hardware queues, and every extra stream makes two of them share one)
        side = _COPY_STREAMS.get(self.device.index)
        if side is None:
            side = _COPY_STREAMS[self.device.index] = torch.cuda.Stream(device=self.device)
        with torch.cuda.stream(side):
            t = buf.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(side)
        main = self.torch_stream()
        main.wait_event(ev)
        t.record_stream(main)
        return t

    def from_host(self, t, tag="h2d"):
        """Host tensor -> device through a pinned staging buffer (asynchronous on the current stream)."""
        buf = self._pinned(t.shape, t.dtype, tag)
        buf.copy_(t)
        return buf.to(self.device, non_blocking=True)

    def synchronize(self):
        self.torch_stream().synchronize()

    def record_event(self, timing=True):
        """HIP event recorded on the stream the kernels are launched on."""
        ev = torch.cuda.Event(enable_timing=timing)
        ev.record(self.torch_stream())
        return ev

    def wait_event(self, ev):
        """Make this object's stream wait (on the GPU) for ``ev``; the host does not block."""
        self.torch_stream().wait_event(ev)

    def side_ops(self):
        """A second ``HipOps`` on the device's SIDE compute stream: work that nothing on the launch stream needs for a while -- the
        message columns of an epoch's irregular candidates, taken while the epoch's first chain of single-work-group kernels
        leaves 255 compute units idle -- is enqueued there between ``side.wait_event(launch-stream event)`` and
        ``self.wait_event(side event)``.  ONE side stream per device for the life of the process (the reasons given at
        ``from_pinned_side``: allocator pools per stream, hardware queues shared between streams); batches in flight share it."""
        idx = self.device.index
        ops = _SIDE_OPS.get(idx)
        if ops is None:
            ops = _SIDE_OPS[idx] = HipOps(self.device, stream=torch.cuda.Stream(device=self.device))
        return ops

    def own_side_ops(self):
        """A side stream of THIS object's own (one per batch in flight: ``SlotPool`` keeps the objects, so the streams are
        created once) -- ``_config.PIPELINED_SIDE_STREAM``."""
        ops = self.__dict__.get("_own_side")
        if ops is None:
            ops = self.__dict__["_own_side"] = HipOps(self.device, stream=torch.cuda.Stream(device=self.device))
        return ops

    def side_context(self):
        """``with ops.side_context():`` -- torch's current stream is the side stream inside the block, so that what is allocated
        there (work buffers of the wrappers included) comes from THAT stream's pool and is recycled in its order."""
        return torch.cuda.stream(self.side_ops().stream)

    def own_side_context(self):
        return torch.cuda.stream(self.own_side_ops().stream)
