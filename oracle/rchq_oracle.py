"""CPU oracle for the kernel-recombination hot path (``BASQ/_rchq.py`` of ma921/BASQ).

TEST INFRASTRUCTURE ONLY.  Nothing under ``basq_amd/`` may import this module;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg use it, and only as the checker / the timed CPU baseline.

Pinning
-------
``tests/test_oracle_vs_reference.py`` (runs only where ``/root/reference`` is
mounted) and ``oracle/make_golden.py`` drive this restatement and the imported
reference ``BASQ._rchq.recombination`` with the same callables, dtype (float64)
and ``torch.manual_seed``; the outputs are required to be **bit-identical**
(indices and weights).  The committed fixtures under ``tests/golden/`` are the
reference's outputs, so the oracle is pinned on boxes without the reference as
well.  (Kernel *values* at the gpytorch boundary are unpinned: see
``oracle/kernels_oracle.py``.)

The floating-point op sequence follows the reference line by line -- the same
torch expressions in the same order -- because the contract is numerical:

================================  =====================================
oracle function                   reference (``BASQ/_rchq.py``)
================================  =====================================
``recombination_oracle``          ``recombination`` :4-25, ``rc_kernel_svd`` :34-40
``nystrom_basis``                 ``ker_svd_sparsify`` :28-31
``divide_and_recombine``          ``Mod_Tchernychova_Lyons`` :43-130
``caratheodory_reduce``           ``Tchernychova_Lyons_CAR`` :133-175
================================  =====================================

On top of the reference's behaviour the oracle records a ``Trace`` (per-round
sizes, block sums, barycentres, survivor sets, the tie margin of every pivot
choice) so tests can compare intermediates of the HIP path, not only the result.
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field

import torch


@dataclass
class RoundTrace:
    remaining: int
    n_blocks: int
    n_tail: int
    bary: torch.Tensor | None = None        # [S, q]  barycentres handed to the reduction
    tot_weights: torch.Tensor | None = None  # [S]
    kept_sets: torch.Tensor | None = None    # ascending set ids that survived
    kept_weights: torch.Tensor | None = None
    tie_margin: float = float("inf")


@dataclass
class Trace:
    rounds: list = field(default_factory=list)
    U: torch.Tensor | None = None
    n_kernel_calls: int = 0
    t_kernel: float = 0.0
    t_car: float = 0.0
    t_basis: float = 0.0
    keep_tensors: bool = True

    @property
    def tie_margin(self) -> float:
        return min([r.tie_margin for r in self.rounds], default=float("inf"))


def nystrom_basis(pt, rank, kernel):
    """``ker_svd_sparsify`` (:28-31): randomised SVD of the Nystrom Gram matrix, ``U = -Uq^T``."""
    Uq, S, _ = torch.svd_lowrank(kernel(pt, pt), q=rank)
    return S, -1 * Uq.T


def caratheodory_reduce(X, mu, trace: RoundTrace | None = None):
    """``Tchernychova_Lyons_CAR`` (:133-175): reduce M weighted points in R^q to <= q+1.

    ``X`` is ``[M, q]``; ``mu`` (``[M]``) is modified in place.  Returns
    ``(w_star, idx_star)``: the surviving positive weights and their positions.
    """
    X = torch.cat([torch.ones(X.size(0)).unsqueeze(0).T, X], dim=1)   # :138
    M, s = X.shape
    _, _, Vh = torch.linalg.svd(X.T)                                   # :140 (full)
    Phi = Vh[-(M - s):, :].T                                           # :143 null space, [M, M-s]
    margin = float("inf")
    for _ in range(M - s):                                             # :146-171
        col = Phi[:, 0]
        pos = col > 0
        alpha = torch.zeros(len(mu))
        alpha[pos] = mu[pos] / col[pos]
        ratios = alpha[pos]
        j = torch.arange(len(mu))[pos][torch.argmin(ratios)]
        if trace is not None and ratios.numel() > 1:
            two = torch.topk(ratios, 2, largest=False).values
            if two[0] > 0:
                margin = min(margin, float((two[1] - two[0]) / two[0]))
        mu[:] = mu - alpha[j] * col
        mu[j] = 0.0
        Phi = Phi[:, 1:]
        Phi = Phi - torch.matmul(Phi[j].unsqueeze(1), col.unsqueeze(1).T).T / col[j]
        Phi[j, :] = 0.0
    if trace is not None:
        trace.tie_margin = margin
    keep = mu > 0
    return mu[keep], torch.arange(M)[keep]


def divide_and_recombine(samp, U, pt, kernel, trace: Trace | None = None):
    """``Mod_Tchernychova_Lyons`` (:43-130).  Returns ``(w_star, idx_star)``.

    NB ``mu`` starts uniform whatever the caller passed (:53 overwrites it).
    """
    N = len(samp)
    q, m = U.shape
    S = 2 * (q + 1)                                                    # :50
    mu = torch.ones(N) / N                                             # :53
    live = torch.arange(N)[mu != 0]                                    # :55-56
    R = len(live)

    def timed_kernel(a, b):
        if trace is None:
            return kernel(a, b)
        t0 = time.perf_counter()
        out = kernel(a, b)
        trace.t_kernel += time.perf_counter() - t0
        trace.n_kernel_calls += 1
        return out

    while True:
        if R <= q + 1:                                                 # :60-63
            sel = torch.arange(len(mu))[mu > 0]
            return mu[sel], sel
        if R <= S:                                                     # :65-74
            F = U @ timed_kernel(pt, samp[live])
            rt = RoundTrace(R, 0, R) if trace is not None else None
            t0 = time.perf_counter()
            w, keep = caratheodory_reduce(F.T, torch.clone(mu[live]), rt)
            if trace is not None:
                trace.t_car += time.perf_counter() - t0
                trace.rounds.append(rt)
            live = live[keep]
            mu[:] = 0.0
            mu[live] = w
            return mu[mu > 0], live

        nb = int(R / S)                                                # :76
        grid = live[: nb * S].reshape(nb, -1)                          # :78  grid[i, s] = live[i*S + s]
        acc = torch.zeros((m, S))                                      # :79
        for i in range(nb):                                            # :81-86  HOT LOOP
            blk = live[i * S:(i + 1) * S]
            acc += torch.multiply(timed_kernel(pt, samp[blk]), mu[blk].unsqueeze(0))
        feat_t = U @ acc                                               # :88
        feat = feat_t.T                                                # :89 (view)
        tot = torch.sum(mu[grid], 0)                                   # :90
        tail = live[nb * S:]                                           # :91
        if len(tail):                                                  # :93-99 ragged tail -> last set
            Ft = U @ timed_kernel(pt, samp[tail])
            feat[-1] += torch.multiply(Ft.T, mu[tail].unsqueeze(1)).sum(axis=0)
            tot[-1] += torch.sum(mu[tail], 0)
        feat = torch.divide(feat, tot.unsqueeze(0).T)                  # :101 barycentres

        rt = None
        if trace is not None:
            rt = RoundTrace(R, nb, len(tail))
            if trace.keep_tensors:
                rt.bary = feat.clone()
                rt.tot_weights = tot.clone()
        t0 = time.perf_counter()
        w, keep = caratheodory_reduce(feat, torch.clone(tot), rt)      # :103-105
        if trace is not None:
            trace.t_car += time.perf_counter() - t0
            rt.kept_sets = keep.clone()
            rt.kept_weights = w.clone()
            trace.rounds.append(rt)

        survivors = grid[:, keep].reshape(-1)                          # :107
        drop = torch.ones(grid.shape[1]).to(torch.bool)
        drop[keep] = 0
        mu[grid[:, drop].reshape(-1)] = 0.0                            # :108-112
        scaled = torch.multiply(mu[grid[:, keep]], w)                  # :113
        scaled = torch.divide(scaled, tot[keep])                       # :114
        mu[survivors] = scaled.reshape(-1)                             # :115

        hit = torch.arange(len(keep))[(keep == S - 1) != 0]            # :117-118
        if len(hit) > 0:                                               # :120-124 last set survived
            t_mu = torch.multiply(mu[tail], w[hit])
            t_mu = torch.divide(t_mu, tot[keep[hit]])
            mu[tail] = t_mu
            survivors = torch.cat([survivors, tail])
        else:                                                          # :125-127
            mu[tail] = 0.0
        live = torch.clone(survivors)                                  # :129-130
        R = len(live)


def recombination_oracle(pts_rec, pts_nys, num_pts, kernel, trace: Trace | None = None):
    """``recombination`` (:4-25) -> ``(idx, w)``; ``init_weights`` has no effect in the reference (:53)."""
    t0 = time.perf_counter()
    _, U = nystrom_basis(pts_nys, num_pts - 1, kernel)                 # :36
    if trace is not None:
        trace.t_basis = time.perf_counter() - t0
        if trace.keep_tensors:
            trace.U = U.clone()
    w, idx = divide_and_recombine(pts_rec, U, pts_nys, kernel, trace)  # :37-39
    return idx, w
