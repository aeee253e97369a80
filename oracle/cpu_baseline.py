"""Bounded-time CPU baseline for ``bench.py`` (the oracle as the reference's CPU PyTorch path).

TEST/BENCH INFRASTRUCTURE ONLY -- never imported by ``basq_amd``.

A full float64 reference batch at the headline size (N=1e6, m=1e4, n=100) is ~30-110 s of CPU work
on 8 cores, too long for a default ``bench.py`` run.  ``sampled_batch_seconds`` runs the oracle's
op sequence (``oracle/rchq_oracle.py`` == ``BASQ/_rchq.py``) **in full** except for the hot loop
(``_rchq.py:81-86``), of which only every ``stride``-th block is executed and timed; the loop time is
scaled by the inverse of the executed fraction.  Everything else -- Gram matrix, ``torch.svd_lowrank``,
every round's projection, SVD and elimination, the re-weighting -- runs and is timed as is.  Skipping
blocks changes the block sums, hence which sets survive, but not the round structure (each round keeps
<= n of 2n sets), so the per-round sizes and the number of rounds are those of a real batch.
"""
from __future__ import annotations

import time

import torch

from .rchq_oracle import caratheodory_reduce, nystrom_basis


def sampled_batch_seconds(pts_rec, pts_nys, num_pts, kernel, stride: int = 8):
    """-> dict(seconds_per_batch, measured_seconds, loop_fraction, n_rounds, kernel_calls_total, kernel_calls_run)."""
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        return _run(pts_rec, pts_nys, num_pts, kernel, max(1, int(stride)))
    finally:
        torch.set_default_dtype(prev)


def _run(samp, pt, num_pts, kernel, stride):
    t_wall = time.perf_counter()
    t0 = time.perf_counter()
    _, U = nystrom_basis(pt, num_pts - 1, kernel)
    t_fixed = time.perf_counter() - t0
    N = len(samp)
    q, m = U.shape
    S = 2 * (q + 1)
    mu = torch.ones(N) / N
    live = torch.arange(N)
    R = N
    t_loop_est = 0.0
    calls_total = calls_run = rounds = 0
    while R > S:
        rounds += 1
        nb = int(R / S)
        grid = live[: nb * S].reshape(nb, -1)
        acc = torch.zeros((m, S))
        t0 = time.perf_counter()
        ran = 0
        for i in range(0, nb, stride):
            blk = live[i * S:(i + 1) * S]
            acc += torch.multiply(kernel(pt, samp[blk]), mu[blk].unsqueeze(0))
            ran += 1
        dt = time.perf_counter() - t0
        t_loop_est += dt * nb / ran
        calls_total += nb
        calls_run += ran
        t0 = time.perf_counter()
        feat = (U @ acc).T
        tot = torch.sum(mu[grid], 0)
        tail = live[nb * S:]
        if len(tail):
            Ft = U @ kernel(pt, samp[tail])
            feat[-1] += torch.multiply(Ft.T, mu[tail].unsqueeze(1)).sum(axis=0)
            tot[-1] += torch.sum(mu[tail], 0)
            calls_total += 1
            calls_run += 1
        feat = torch.divide(feat, tot.unsqueeze(0).T)
        w, keep = caratheodory_reduce(feat, torch.clone(tot))
        survivors = grid[:, keep].reshape(-1)
        drop = torch.ones(grid.shape[1]).to(torch.bool)
        drop[keep] = 0
        mu[grid[:, drop].reshape(-1)] = 0.0
        mu[survivors] = torch.divide(torch.multiply(mu[grid[:, keep]], w), tot[keep]).reshape(-1)
        if len(keep) and int(keep[-1]) == S - 1 and len(tail):
            mu[tail] = mu[tail] * w[-1] / tot[S - 1]
            survivors = torch.cat([survivors, tail])
        else:
            mu[tail] = 0.0
        live = survivors
        R = len(live)
        t_fixed += time.perf_counter() - t0
    if R > q + 1:
        t0 = time.perf_counter()
        F = U @ kernel(pt, samp[live])
        caratheodory_reduce(F.T, torch.clone(mu[live]))
        t_fixed += time.perf_counter() - t0
        rounds += 1
        calls_total += 1
        calls_run += 1
    return dict(
        seconds_per_batch=t_fixed + t_loop_est,
        measured_seconds=time.perf_counter() - t_wall,
        loop_fraction=calls_run / max(calls_total, 1),
        n_rounds=rounds,
        kernel_calls_total=calls_total,
        kernel_calls_run=calls_run,
    )
