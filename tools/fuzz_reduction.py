"""Randomised stress of the hand-synchronised single-work-group kernels (null space, elimination, Cholesky).

    python tools/fuzz_reduction.py [--cases 150] [--seed 0]

For random shapes (all dispatch paths): the GPU null space must match LAPACK's rows, the elimination must match the CPU
restatement bit for bit, repeated launches on the same input must be bitwise identical (a race would show up as a
flaky last bit), and chol_inv must orthonormalise.  Prints one summary line; exits non-zero on the first failure.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from basq_amd._ops import HipOps              # noqa: E402
from tests.cpu_stand_in import CpuStandInOps  # noqa: E402


def run(cases, seed, ops=None):
    ops, cpu = ops or HipOps(torch.device("cuda:0")), CpuStandInOps()
    g = torch.Generator().manual_seed(seed)
    worst_ns, n_elim, n_chol = 0.0, 0, 0
    for case in range(cases):
        s = int(torch.randint(1, 131, (1,), generator=g))
        M = int(torch.randint(s + 1, min(2 * s + 3, 300) + 1, (1,), generator=g))
        X = torch.randn(s, M, generator=g, dtype=torch.float64)
        X[0] = 1.0
        if case % 7 == 0:                                   # badly scaled rows, like real barycentre matrices
            X[1:] *= torch.logspace(0, -6, s - 1, dtype=torch.float64).unsqueeze(1) if s > 1 else 1.0
        ref = torch.linalg.svd(X)[2][s:]
        Xd = ops.to_device(X)
        P1 = ops.nullspace(Xd, s, M)
        P2 = ops.nullspace(Xd, s, M)
        assert torch.equal(P1, P2), f"nullspace not repeatable at s={s} M={M}"
        err = (P1.cpu() - ref).abs().max().item()
        scale = 1.0 if case % 7 else 1e4                    # ill-conditioned cases: looser (still rounding-level)
        assert err <= 1e-10 * scale, f"nullspace off by {err:.2e} at s={s} M={M}"
        worst_ns = max(worst_ns, err / scale)
        mu = torch.rand(M, generator=g, dtype=torch.float64) + 0.05
        mu /= mu.sum()
        kr_c, kept_c, w_c, info_c = cpu.car_eliminate(ref.clone().contiguous(), mu.clone(), M, s)
        outs = []
        for _ in range(2):
            kr, kept, w, info = ops.car_eliminate(ops.to_device(ref.clone().contiguous()), ops.to_device(mu.clone()), M, s)
            outs.append((kr.cpu(), kept.cpu(), w.cpu(), info.cpu()))
        nk = int(info_c[0])
        for kr, kept, w, info in outs:
            assert info.tolist() == info_c.tolist() and torch.equal(kr, kr_c), f"elimination differs at s={s} M={M}"
            assert torch.equal(kept[:nk], kept_c[:nk]) and torch.equal(w[:nk], w_c[:nk]), f"weights differ at s={s} M={M}"
        n_elim += 1
        if case % 3 == 0:
            q = int(torch.randint(1, 210, (1,), generator=g))
            Y = torch.randn(3 * q + 5, q, generator=g, dtype=torch.float64)
            G = Y.T @ Y
            W1, i1 = ops.chol_inv(ops.to_device(G.clone()))
            W2, i2 = ops.chol_inv(ops.to_device(G.clone()))
            assert int(i1.cpu()[0]) == 0 and torch.equal(W1, W2), f"chol_inv flaky at q={q}"
            Q = Y @ W1.cpu()
            assert (Q.T @ Q - torch.eye(q, dtype=torch.float64)).abs().max().item() <= 1e-9, f"chol_inv wrong at q={q}"
            n_chol += 1
    return (f"fuzz ok: {cases} shapes, worst |nullspace - LAPACK| {worst_ns:.2e}, {n_elim} eliminations bit-exact and "
            f"repeatable, {n_chol} Cholesky factorisations")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=150)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    print(run(a.cases, a.seed))


if __name__ == "__main__":
    main()
