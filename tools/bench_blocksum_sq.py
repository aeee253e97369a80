"""Time basq_blocksum_sq_f64 / basq_blocksum_sq_geo_f64 in isolation (config 5m's shapes: m = 5000, n_obs = 202, d = 10, S = 200).

    [BASQ_HIP_LIB=variant.so] python tools/bench_blocksum_sq.py [--R 100000] [--reps 5]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from basq_amd._ops import HipOps                                  # noqa: E402
from basq_amd.kernels import StationaryKernel                      # noqa: E402
from basq_amd.pools import gmm_pool                                # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--R", type=int, default=100_000)
    ap.add_argument("--m", type=int, default=5_000)
    ap.add_argument("--n-obs", type=int, default=202)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    d, S, C = 10, 200, 16
    ops = HipOps(torch.device("cuda:0"))
    spec = StationaryKernel("rbf", 2.0, 1.0).spec(d)
    pts = gmm_pool(a.R + a.m + a.n_obs, d, 3).to(ops.device)
    nys, obs, cand = pts[:a.m], pts[a.m:a.m + a.n_obs], pts[a.m + a.n_obs:]
    cen = ops.col_mean(nys)
    pa = ops.pack(spec, torch.cat([nys, obs], 0).contiguous(), cen, 0, pad_rows_to=64)
    pb = ops.pack(spec, cand.contiguous(), cen, 1)
    mu = ops.zeros(a.R) + 1.0 / a.R
    n4, mp = (a.n_obs + 3) // 4 * 4, (a.m + 63) // 64 * 64
    bT = ops.zeros(n4, mp)
    bT[:a.n_obs, :a.m] = 0.01 * torch.randn(a.n_obs, a.m, dtype=torch.float64, device=ops.device)
    kobs = ops.zeros(n4, a.R)
    ops.gram_into(spec, pa[a.m:a.m + a.n_obs], a.n_obs, pb, a.R, kobs)
    nb = a.R // S
    reg = (nb // C) * C * S
    geo = ops.geo_init(2, a.R, S, reg, 0, a.R)

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.reps

    flop = float(reg) * a.m * (2 * a.n_obs + 3 * d + 9)
    for name, fn in (("host geometry, 16 classes", lambda: ops.blocksum_sq(spec, pa, a.m, pb, mu, reg, 0, nb * S, S, C, bT, kobs, a.n_obs, 0.0, class_mod=C)),
                     ("descriptor,    16 classes", lambda: ops.blocksum_sq_geo(spec, pa, a.m, pb, mu, geo[0], 1, S, C, bT, kobs, a.n_obs, 0.0, class_mod=C))):
        ms = timed(fn)
        print(f"blocksum_sq {name}: {ms:8.3f} ms for {reg} candidates x {a.m} rows x {a.n_obs} observations = {flop / ms / 1e9:6.1f} TF/s "
              f"({flop / ms / 1e9 / 78.6:.2f} of the nominal fp64 peak)")


if __name__ == "__main__":
    main()
