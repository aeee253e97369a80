"""Where the HOST spends its time in a headline batch (cProfile of a few batches on one GPU).

    python tools/host_profile.py [--batches 5]
"""
import argparse
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import basq_amd                                   # noqa: E402

basq_amd.configure_hw_queues()                                      # (before the first GPU call)
from basq_amd.pools import gmm_pool               # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=5)
    ap.add_argument("--N", type=int, default=1_000_000)
    ap.add_argument("--many", type=int, default=0, help="profile recombination_many with this many batches in flight instead")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    pool = gmm_pool(a.N, 10, 0).to(dev)
    nys = pool[:a.N // 100].contiguous()
    kern = basq_amd.kernels.StationaryKernel("rbf", 2.0, 1.0)
    for _ in range(2):
        torch.manual_seed(1)
        basq_amd.recombination(pool, nys, 100, kern, dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.batches):
        torch.manual_seed(1)
        basq_amd.recombination(pool, nys, 100, kern, dev)
    torch.cuda.synchronize()
    print(f"unprofiled: {(time.perf_counter() - t0) / a.batches * 1e3:.2f} ms/batch")
    if a.many:
        calls = [(pool, nys, 100, kern)] * a.batches
        basq_amd.recombination_many(calls, dev, in_flight=a.many, seeds=[1] * a.batches)
        torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    t0 = time.perf_counter()
    if a.many:
        basq_amd.recombination_many(calls, dev, in_flight=a.many, seeds=[1] * a.batches)
    else:
        for _ in range(a.batches):
            torch.manual_seed(1)
            basq_amd.recombination(pool, nys, 100, kern, dev)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    pr.disable()
    print(f"profiled:   {dt / a.batches * 1e3:.2f} ms/batch (cProfile overhead included)")
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)


if __name__ == "__main__":
    main()
