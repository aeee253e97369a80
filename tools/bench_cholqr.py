"""Time the CholeskyQR kernels in isolation: fused (basq_cholqr_f64) against its two halves (basq_chol_factor_f64, basq_trsm_rows_f64).

    python tools/bench_cholqr.py [--rows 10000] [--q 99] [--reps 100]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from basq_amd._ops import HipOps  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000)
    ap.add_argument("--q", type=int, default=99)
    ap.add_argument("--reps", type=int, default=100)
    a = ap.parse_args()
    ops = HipOps(torch.device("cuda:0"))
    g = torch.Generator().manual_seed(0)
    X = ops.to_device(torch.randn(a.rows, a.q, generator=g, dtype=torch.float64))
    G0 = X.t() @ X

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
        return 1e3 * e0.elapsed_time(e1) / a.reps

    Gs = [G0.clone() for _ in range(a.reps + 3)]
    it = iter(Gs)
    t_clone = timed(lambda: G0.clone())
    t_fused = timed(lambda: ops.cholqr(G0.clone(), X)) - t_clone
    t_factor = timed(lambda: ops.chol_factor(G0.clone())) - t_clone
    L = G0.clone()
    ops.chol_factor(L)
    t_trsm = timed(lambda: ops.trsm_rows(X, L))
    Q, _ = ops.cholqr(G0.clone(), X)
    err = (Q.t() @ Q - torch.eye(a.q, dtype=torch.float64, device=Q.device)).abs().max().item()
    print(f"rows {a.rows} q {a.q}: fused {t_fused:7.1f} us   factor alone {t_factor:7.1f} us   solve alone {t_trsm:7.1f} us   "
          f"(clone {t_clone:.1f} us subtracted)   |Q^T Q - I| = {err:.1e}")


if __name__ == "__main__":
    main()
