"""Time the single-work-group kernels in isolation: basq_nullspace_f64, basq_car_eliminate_f64, basq_chol_inv_f64.

    python tools/bench_reduction.py [s M] [--reps 50]

Prints the mean kernel time (HIP events on the launch stream) for the [s, M] Caratheodory matrix of a round
(default 100 x 200, the headline config) and checks the null space against host LAPACK.
"""
import argparse
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from basq_amd._ops import HipOps  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("shape", nargs="*", type=int, default=[100, 200])
    ap.add_argument("--reps", type=int, default=50)
    a = ap.parse_args()
    s, M = a.shape
    ops = HipOps(torch.device("cuda:0"))
    g = torch.Generator().manual_seed(0)
    X = torch.randn(s, M, generator=g, dtype=torch.float64)
    X[0] = 1.0
    mu = torch.rand(M, generator=g, dtype=torch.float64) + 0.05
    mu /= mu.sum()
    Xd, mud = ops.to_device(X), ops.to_device(mu)
    ref = torch.linalg.svd(X)[2][s:]
    P = ops.nullspace(Xd, s, M)
    print(f"[{s} x {M}] max |nullspace - LAPACK rows| = {(P.cpu() - ref).abs().max().item():.2e}")

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.reps * 1e3

    print(f"nullspace      {timed(lambda: ops.nullspace(Xd, s, M)):9.1f} us / call")
    print(f"car_eliminate  {timed(lambda: ops.car_eliminate(P.clone(), mud.clone(), M, s)):9.1f} us / call (incl. 2 clones)")
    q = s - 1 if s > 1 else 1
    Y = torch.randn(4 * q + 7, q, generator=g, dtype=torch.float64)
    G = ops.to_device(Y.T @ Y)
    print(f"chol_inv q={q:<4d}{timed(lambda: ops.chol_inv(G.clone())):9.1f} us / call (incl. 1 clone)")
    print(f"chol_factor q={q:<4d}{timed(lambda: ops.chol_factor(G.clone())):6.1f} us / call (incl. 1 clone)")
    Xt = ops.to_device(torch.randn(10000, q, generator=g, dtype=torch.float64))
    Lf = G.clone()
    ops.chol_factor(Lf)
    print(f"trsm_rows q={q:<4d}{timed(lambda: ops.trsm_rows(Xt, Lf)):8.1f} us / call ([10000, q])")


if __name__ == "__main__":
    main()
