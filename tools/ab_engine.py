"""Interleaved A/B timing of engine switches on one box (boxes of the pool differ by ~10-20 % in host speed, so only
same-process comparisons mean anything).

    python tools/ab_engine.py LATE_CHUNKS 0 1 [--reps 8]

Sets ``basq_amd._config.<NAME>`` to each value in turn (A B A B ...), times ``reps`` headline batches per visit and
prints the mean / min per value.
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import basq_amd                                     # noqa: E402
import basq_amd._config as eng                      # noqa: E402
from basq_amd.pools import gmm_pool                 # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("name")
    ap.add_argument("values", nargs="+")
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--visits", type=int, default=3)
    ap.add_argument("--module", default="_config", help="basq_amd submodule holding the switch (_config, _partition, _ops)")
    a = ap.parse_args()
    global eng
    import importlib

    eng = importlib.import_module("basq_amd." + a.module)
    cur = getattr(eng, a.name)
    vals = [type(cur)(int(v)) if isinstance(cur, (bool, int)) else type(cur)(v) for v in a.values]
    dev = torch.device("cuda:0")
    N, d, n = 1_000_000, 10, 100
    pts = gmm_pool(N, d, 0).to(dev)
    nys = pts[: N // 100].contiguous()
    kern = basq_amd.kernels.StationaryKernel("rbf", 2.0)

    def batch():
        torch.manual_seed(1)
        return basq_amd.recombination(pts, nys, n, kern, dev)

    for _ in range(2):
        batch()
    torch.cuda.synchronize()
    times = {v: [] for v in vals}
    ref = None
    for _ in range(a.visits):
        for v in vals:
            setattr(eng, a.name, v)
            batch()
            torch.cuda.synchronize()
            for _ in range(a.reps):
                t0 = time.perf_counter()
                idx, w = batch()
                torch.cuda.synchronize()
                times[v].append((time.perf_counter() - t0) * 1e3)
            if ref is None:
                ref = (idx.clone(), w.clone())
            assert torch.equal(idx, ref[0]) and torch.allclose(w, ref[1], rtol=1e-9, atol=0), "the switch changed the result"
    for v in vals:
        t = sorted(times[v])
        print(f"{a.name}={v}: mean {sum(t) / len(t):7.2f} ms  median {t[len(t) // 2]:7.2f}  min {t[0]:7.2f}  ({len(t)} batches)")


if __name__ == "__main__":
    main()
