"""Time the BASELINE.json configurations 2-5 (tests/cases.py: cfg2..cfg5) on one GPU, with the engine's phase breakdown.

    python tools/bench_configs.py [--reps 3] [--only cfg4_matern52_1e6_d32]

These are parity-test cases, not bench lines (bench.py reports config 3); this tool only shows where each one spends
its time.  Pools are generated on the host and moved to HBM before the clock.
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import basq_amd                                                    # noqa: E402
from tests.cases import BY_NAME, build_pool, build_product_kernel  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    names = [a.only] if a.only else ["cfg2_rbf_1e5", "cfg3_rbf_1e6", "cfg4_matern52_1e6_d32", "cfg5_wsabil_5e5",
                                     "cfg5m_wsabim_5e5"]
    for name in names:
        c = BY_NAME[name]
        pts, nys = build_pool(c)
        pts, nys = pts.to(dev), nys.to(dev)
        kern = build_product_kernel(c)
        for _ in range(2):
            torch.manual_seed(c["torch_seed"])
            basq_amd.recombination(pts, nys, c["n"], kern, dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            torch.manual_seed(c["torch_seed"])
            idx, w = basq_amd.recombination(pts, nys, c["n"], kern, dev)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / a.reps * 1e3
        tr = basq_amd.EngineTrace(host_sync=True)
        torch.manual_seed(c["torch_seed"])
        basq_amd.recombination(pts, nys, c["n"], kern, dev, trace=tr)
        keys = ["basis", "blocksum", "project", "nullspace", "host_svd", "eliminate", "compact", "wsabim_sq"]
        parts = "  ".join(f"{k} {tr.timers[k] * 1e3:.1f}" for k in keys if k in tr.timers)
        print(f"{name:24s} N={c['N']:<8d} d={c['d']:<3d} n={c['n']:<4d} m={c['m']:<6d} {ms:8.1f} ms/batch "
              f"({len(tr.rounds)} rounds, {len(idx)} selected) | synchronised phases (ms): {parts}")


if __name__ == "__main__":
    main()
