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

SURVEY §8d asks for the matrix {float64, float32} x {8 threads, all host cores}: ``baseline_matrix``.
The hot loop is thousands of ``[m, 2n]`` kernel blocks: with every core of a many-core host in the MKL/OpenMP
team they thrash (round 1 recorded 547 s vs 31 s for the same batch), so the thread count is part of the
result and the faster setting is the one reported.  ``python -m oracle.cpu_baseline --full`` runs ONE
un-sampled batch (``stride=1``) to anchor the extrapolation (``profiles/r02_cpu_full_batch.txt``).
"""
from __future__ import annotations

import time

import torch

from .rchq_oracle import caratheodory_reduce, nystrom_basis


def sampled_batch_seconds(pts_rec, pts_nys, num_pts, kernel, stride: int = 8, dtype=torch.float64,
                          threads: int | None = None, loop_budget_s: float | None = None,
                          give_up_above_s: float | None = None):
    """-> dict(seconds_per_batch, measured_seconds, loop_fraction, n_rounds, kernel_calls_total, kernel_calls_run,
    threads, dtype).  ``threads``: ``torch.set_num_threads`` for the duration of the call (None = leave as is)."""
    prev = torch.get_default_dtype()
    prev_threads = torch.get_num_threads()
    torch.set_default_dtype(dtype)
    if threads:
        torch.set_num_threads(int(threads))
    try:
        # warm-up at this dtype / team size (first-use costs of MKL's thread team and LAPACK workspaces: ~1 s otherwise
        # lands in the Gram + svd_lowrank timer of the first cell)
        nw = min(len(pts_rec), 1000)
        _run(pts_rec[:nw].to(dtype), pts_nys[:max(1, min(len(pts_nys), nw // 20))].to(dtype), num_pts, kernel, 1, None, None)
        res = _run(pts_rec.to(dtype), pts_nys.to(dtype), num_pts, kernel, max(1, int(stride)), loop_budget_s, give_up_above_s)
        res["threads"] = torch.get_num_threads()
        res["dtype"] = str(dtype).replace("torch.", "")
        return res
    finally:
        torch.set_default_dtype(prev)
        if threads:
            torch.set_num_threads(prev_threads)


def baseline_matrix(pts_rec, pts_nys, num_pts, kernel, stride: int, seed: int = 1, thread_counts=None,
                    dtypes=(torch.float64, torch.float32), loop_budget_s: float | None = 4.0):
    """every (dtype, thread count) cell, same seed before each.  -> list of result dicts."""
    import os

    ncpu = os.cpu_count() or 1
    if thread_counts is None:
        thread_counts = sorted({min(8, ncpu), ncpu})
    out = []
    for dt in dtypes:
        best = None                                            # thread counts ascending: the 8-thread cell sets the bar
        for th in thread_counts:
            torch.manual_seed(seed)
            res = sampled_batch_seconds(pts_rec, pts_nys, num_pts, kernel, stride, dt, th, loop_budget_s,
                                        give_up_above_s=None if best is None else 5.0 * best)
            best = res["seconds_per_batch"] if best is None else min(best, res["seconds_per_batch"])
            out.append(res)
    return out


def _run(samp, pt, num_pts, kernel, stride, loop_budget_s, give_up_above_s=None):
    t_wall = time.perf_counter()
    t0 = time.perf_counter()
    _, U = nystrom_basis(pt, num_pts - 1, kernel)
    t_basis = time.perf_counter() - t0
    t_fixed = t_basis
    N = len(samp)
    q, m = U.shape
    S = 2 * (q + 1)
    mu = torch.ones(N) / N
    live = torch.arange(N)
    R = N
    t_loop_est = t_loop_run = 0.0
    calls_total = calls_run = rounds = 0
    nb_first = max(1, int(N / S))
    while R > S:
        rounds += 1
        nb = int(R / S)
        # this round's share of the loop budget (the block counts halve every round: sum ~ 2 x the first round's)
        round_budget = None if loop_budget_s is None else loop_budget_s * nb / (2.0 * nb_first)
        grid = live[: nb * S].reshape(nb, -1)
        acc = torch.zeros((m, S))
        t0 = time.perf_counter()
        ran = 0
        for i in range(0, nb, stride):
            blk = live[i * S:(i + 1) * S]
            acc += torch.multiply(kernel(pt, samp[blk]), mu[blk].unsqueeze(0))
            ran += 1
            if give_up_above_s is not None and rounds == 1 and ran == 2:
                projected = t_basis + (time.perf_counter() - t0) / 2.0 * (2.0 * nb_first)
                if projected > give_up_above_s:
                    return dict(seconds_per_batch=projected, measured_seconds=time.perf_counter() - t_wall,
                                basis_seconds=t_basis, fixed_seconds=t_basis, loop_seconds_run=time.perf_counter() - t0,
                                loop_seconds_scaled=projected - t_basis, loop_fraction=2.0 / (2.0 * nb_first), n_rounds=0,
                                kernel_calls_total=int(2 * nb_first), kernel_calls_run=2, abbreviated=True)
            if round_budget is not None and time.perf_counter() - t0 > round_budget:   # >= 1 block per round always runs
                break
        dt = time.perf_counter() - t0
        t_loop_run += dt
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
        basis_seconds=t_basis,
        fixed_seconds=t_fixed,
        loop_seconds_run=t_loop_run,
        loop_seconds_scaled=t_loop_est,
        loop_fraction=calls_run / max(calls_total, 1),
        n_rounds=rounds,
        kernel_calls_total=calls_total,
        kernel_calls_run=calls_run,
    )


def main():
    """``python -m oracle.cpu_baseline [--full] [--threads 8] [--dtype float64] [--stride K]``: one cell, JSON to stdout.

    ``--full`` = stride 1: the whole batch is executed (the anchor of the sampled estimate)."""
    import argparse
    import json
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from basq_amd.pools import gmm_pool          # seeded pool generator (pure torch, no GPU)
    from oracle.kernels_oracle import StationaryOracle

    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=1_000_000)
    ap.add_argument("--d", type=int, default=10)
    ap.add_argument("--n", type=int, default=100)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--dtype", default="float64", choices=["float64", "float32"])
    ap.add_argument("--stride", type=int, default=20)
    ap.add_argument("--full", action="store_true")
    ap.add_argument("--pool-seed", type=int, default=0)
    a = ap.parse_args()
    pool = gmm_pool(a.N, a.d, a.pool_seed)
    m = int(a.N * 1e-2)
    torch.manual_seed(1)
    res = sampled_batch_seconds(pool, pool[:m], a.n, StationaryOracle("rbf", 2.0, 1.0), 1 if a.full else a.stride,
                                getattr(torch, a.dtype), a.threads, None if a.full else 8.0)
    res.update(N=a.N, d=a.d, n=a.n, m=m, host_cores=os.cpu_count(), stride=1 if a.full else a.stride)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
