"""Golden vectors for the SOBER-flavoured entry: runs the REFERENCE ``/root/reference/SOBER/_rchq.py`` (float64).
``python -m oracle.make_golden_sober``"""
import json
import os
import sys

import numpy as np
import torch

CASES = [
    dict(name="sober_uniform_1e4", N=10_000, d=10, m=100, n=100, family="rbf", lengthscale=2.0, pool_seed=3, weights="none"),
    dict(name="sober_is_ragged", N=12_345, d=7, m=123, n=37, family="rbf", lengthscale=1.5, pool_seed=3, weights="is"),
    dict(name="sober_zeros_matern", N=5_000, d=3, m=64, n=30, family="matern52", lengthscale=2.0, pool_seed=3, weights="zeros"),
    dict(name="sober_tiny_final", N=150, d=3, m=100, n=100, family="rbf", lengthscale=2.0, pool_seed=5, weights="is"),
    # with an objective row (calc_obj, SOBER/_rchq.py:14, :67-69, :78-104).  Only the single-reduction branch
    # (N <= 2n) can be pinned: for larger pools the reference itself raises at :140-142 (a [S,1] sum is added in
    # place to a [1,S] buffer), see tests/test_sober.py::test_reference_objective_branch_fails_for_large_pools.
    dict(name="sober_obj_tiny_final", N=150, d=3, m=100, n=100, family="rbf", lengthscale=2.0, pool_seed=5, weights="is",
         objective="bump"),
]


def case_objective(c):
    """``calc_obj`` of the case, or None.  Only correctly rounded IEEE operations (no libm): bit-reproducible."""
    if c.get("objective", "none") == "none":
        return None

    def bump(X):
        d = X.shape[1]
        return 1.0 / (1.0 + (X * X).sum(1) / d) + 0.125 * X[:, 0]

    return bump


def case_weights(c):
    """Importance weights from an integer stream (bit-reproducible across hosts)."""
    if c["weights"] == "none":
        return None
    rng = np.random.Generator(np.random.PCG64(1000 + c["N"]))
    k = rng.integers(1, 1 << 20, size=c["N"], dtype=np.int64).astype(np.float64)
    if c["weights"] == "zeros":
        k[rng.integers(0, 10, size=c["N"]) < 3] = 0.0
    return torch.from_numpy(k / k.sum())


def main():
    import warnings

    sys.dont_write_bytecode = True
    sys.path.insert(0, "/root/reference")
    from SOBER._rchq import recombination as sober_recombination

    from basq_amd.pools import gmm_pool, pool_digest
    from oracle.kernels_oracle import StationaryOracle

    torch.set_default_dtype(torch.float64)
    out = []
    for c in CASES:
        pts = gmm_pool(c["N"], c["d"], c["pool_seed"])
        w0 = case_weights(c)
        k = StationaryOracle(c["family"], c["lengthscale"], 1.0)
        torch.manual_seed(1)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            idx, w = sober_recombination(pts, pts[:c["m"]], c["n"], k, torch.device("cpu"), torch.float64,
                                         init_weights=None if w0 is None else w0.clone(), calc_obj=case_objective(c))
        out.append(dict(case=c, pool_digest=pool_digest(pts), idx=[int(v) for v in idx], w=[float(v) for v in w]))
        print(c["name"], len(idx), float(w.sum()))
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "sober.json")
    json.dump(out, open(path, "w"), indent=0)
    print("wrote", path)


if __name__ == "__main__":
    main()
