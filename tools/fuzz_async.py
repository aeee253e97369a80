"""Differential fuzz on the GPU: descriptor-driven rounds (no host wait per round) against the round-by-round loop, over
random pool sizes (ragged tails at every level), dimensions, batch sizes and kernels (stationary, posterior with noise,
WSABI-L).

    python tools/fuzz_async.py [seed] [cases]

Both paths run the same kernels; what differs is who knows the geometry (device descriptor vs host), rarely the number of residue
classes chosen from a lower bound of the block count, and -- round 6 -- the candidates outside the residue classes inside an epoch
(message columns regrouped vs evaluated afresh): indices must be identical and weights agree to rounding times the problem's own
amplification; a case beyond 1e-9 is printed with both paths' distance from the oracle.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import basq_amd                                    # noqa: E402
import basq_amd._config as eng                     # noqa: E402
from tests.cases import K, build_pool, build_product_kernel, case   # noqa: E402

g = torch.Generator().manual_seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda:0")
bad = 0
for i in range(ncase):
    N = int(torch.randint(2_000, 400_000, (1,), generator=g))
    d = int(torch.randint(2, 14, (1,), generator=g))
    n = int(torch.randint(4, 120, (1,), generator=g))
    m = int(torch.randint(max(n, 20), 1200, (1,), generator=g))
    fam = ["rbf", "matern52", "matern32"][i % 3]
    post = dict(n_obs=int(torch.randint(10, 150, (1,), generator=g)), noise=[1e-10, 1e-6, 1e-3][i % 3], obs_seed=70 + i)
    kern = [K(fam, 1.5 + 0.5 * (i % 3), 1.1), K(fam, 2.0, 1.2, posterior=post), K("rbf", 2.0, 1.0, posterior=post, warp="wsabil")][i % 3 if i % 5 else 0]
    c = case(f"fa{i}", N, d, m, n, kern, pool_seed=900 + i, torch_seed=i)
    pts, nys = build_pool(c)
    pts_d, nys_d = pts.to(dev), nys.to(dev)
    out = []
    for flag in (True, False):
        eng.ASYNC_ROUNDS = flag
        torch.manual_seed(i)
        idx, w = basq_amd.recombination(pts_d, nys_d, n, build_product_kernel(c), dev)
        out.append((idx.cpu(), w.cpu()))
    eng.ASYNC_ROUNDS = True
    (ia, wa), (ib, wb) = out
    same = torch.equal(ia, ib)
    rel = ((wa - wb).abs() / wb.abs()).max().item() if same and len(wb) else float("nan")
    if not same or not rel <= 1e-9:
        bad += 1
        # since round 6 the two paths no longer share every rounding (inside an epoch the descriptor-driven rounds regroup message
        # columns where the round-by-round loop evaluates the irregular candidates afresh): which of them is closer to the oracle?
        from oracle.rchq_oracle import recombination_oracle
        from tests.cases import build_oracle_kernel

        prev = torch.get_default_dtype()
        torch.set_default_dtype(torch.float64)
        torch.manual_seed(i)
        io, wo = recombination_oracle(pts, nys, n, build_oracle_kernel(c)[0])
        torch.set_default_dtype(prev)
        vs = [((w_ - wo).abs() / wo.abs()).max().item() if torch.equal(i_, io) else float("nan") for i_, w_ in out]
        print(f"MISMATCH case {i}: N={N} d={d} n={n} m={m} kernel={kern} same_idx={same} rel={rel}; against the oracle: "
              f"descriptor-driven {vs[0]:.2e}, round by round {vs[1]:.2e}")
print(f"{ncase} cases, {bad} mismatches")
