"""Experiment: do two independent recombination batches on two HIP streams overlap on one MI355X?

    python tools/exp_concurrent.py [--threads 2] [--batches 12] [--N 1000000] [--launches]

Each of ``--threads`` Python threads runs ``--batches`` headline batches on its OWN stream (``torch.cuda.stream``); the
wide kernels of one batch should then fill the 255 CUs the single-work-group reductions of the other leave idle.  The
global CPU generator is shared (selection differs from the sequential run; the WORK is the same) -- a throughput probe,
not a parity run.  ``--launches`` prints every block-sum launch of one traced batch (pairs, microseconds, TF/s).
"""
import argparse
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import basq_amd                                  # noqa: E402
from basq_amd.pools import gmm_pool              # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=2)
    ap.add_argument("--batches", type=int, default=12)
    ap.add_argument("--N", type=int, default=1_000_000)
    ap.add_argument("--d", type=int, default=10)
    ap.add_argument("--n", type=int, default=100)
    ap.add_argument("--launches", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    m = a.N // 100
    kern = basq_amd.kernels.StationaryKernel("rbf", 2.0, 1.0)
    pools = []
    for sd in range(2):
        p = gmm_pool(a.N, a.d, sd)
        pools.append((p[:m].to(dev), p.to(dev)))
    torch.cuda.synchronize()

    def run(k):
        nys, pts = pools[k % len(pools)]
        return basq_amd.recombination(pts, nys, a.n, kern, dev)

    for k in range(3):
        torch.manual_seed(1)
        run(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(a.batches):
        torch.manual_seed(1)
        run(k)
    torch.cuda.synchronize()
    seq = (time.perf_counter() - t0) / a.batches
    print(f"sequential: {seq * 1e3:.2f} ms/batch = {1 / seq:.1f} batches/s", flush=True)

    if a.launches:
        tr = basq_amd.EngineTrace(time_kernels=True, host_sync=False)
        torch.manual_seed(1)
        basq_amd.recombination(pools[0][1], pools[0][0], a.n, kern, dev, trace=tr)
        torch.cuda.synchronize()
        tot = 0.0
        for e0, e1, info in tr.kernel_events:
            us = e0.elapsed_time(e1) * 1e3
            tot += us
            print(f"  blocksum launch: R={info['R']:>10.0f} chunks={info['chunks']:>3d} pairs={info['pairs']:.3e} "
                  f"{us:9.1f} us  {info['pairs'] * 33 / us / 1e6:6.2f} TF/s")
        print(f"  total {tot / 1e3:.3f} ms over {len(tr.kernel_events)} launches; rounds: "
              f"{[r['R'] for r in tr.rounds]}", flush=True)

    for T in sorted({1, a.threads}):
        streams = [torch.cuda.Stream(device=dev) for _ in range(T)]
        lat = [[] for _ in range(T)]

        def worker(t):
            with torch.cuda.stream(streams[t]):
                for k in range(a.batches):
                    t1 = time.perf_counter()
                    run(k + t)
                    lat[t].append(time.perf_counter() - t1)
                streams[t].synchronize()

        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ths = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        n = T * a.batches
        ml = sorted(sum(lat, []))[len(sum(lat, [])) // 2]
        print(f"{T} thread(s) x {a.batches} batches: {dt / n * 1e3:.2f} ms/batch aggregate = {n / dt:.1f} batches/s; "
              f"median latency {ml * 1e3:.2f} ms", flush=True)


if __name__ == "__main__":
    main()
