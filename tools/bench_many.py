"""Headline batches with several in flight (``basq_amd.recombination_many``) against one after the other.

    python tools/bench_many.py [--batches 16] [--inflight 2,3,4] [--N 1000000]

Prints sequential and pipelined throughput and the per-batch latency (host clock from a batch's first launch to its
result) -- the figures bench.py reports as ``value_concurrent2``.  Every pipelined result is checked bit for bit against
its sequential run.
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import basq_amd                                  # noqa: E402

basq_amd.configure_hw_queues()                                      # (before the first GPU call)
from basq_amd._engine import Job, LocalComm, RecombinationEngine   # noqa: E402
from basq_amd._rchq import _DEFAULT_POOL                # noqa: E402
from basq_amd.pools import gmm_pool              # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=16)
    ap.add_argument("--inflight", default="2,3,4")
    ap.add_argument("--N", type=int, default=1_000_000)
    ap.add_argument("--d", type=int, default=10)
    ap.add_argument("--n", type=int, default=100)
    ap.add_argument("--case", default="", help="a tests/cases.py case (e.g. cfg4_matern52_1e6_d32) instead of the headline workload")
    ap.add_argument("--pipelined-only", action="store_true", help="no sequential reference runs (profiles of the pipeline alone)")
    ap.add_argument("--set", action="append", default=[], help="NAME=VALUE for basq_amd._config (e.g. PIPELINE_GATE=0)")
    a = ap.parse_args()
    import basq_amd._config as cfg
    for kv in a.set:
        k, v = kv.split("=")
        setattr(cfg, k, type(getattr(cfg, k))(int(v)))
        print(f"config: {k} = {getattr(cfg, k)}")
    dev = torch.device("cuda", 0)
    if a.case:
        from tests.cases import BY_NAME, build_pool, build_product_kernel

        c = BY_NAME[a.case]
        pts, nys = build_pool(c)
        pts, nys = pts.to(dev), nys.to(dev)
        kern = build_product_kernel(c)
        calls = [(pts, nys, c["n"], kern) for _ in range(a.batches)]
        seeds = [c["torch_seed"]] * a.batches
        print(f"case {a.case}: N={c['N']} d={c['d']} n={c['n']} m={c['m']}")
    else:
        m = a.N // 100
        kern = basq_amd.kernels.StationaryKernel("rbf", 2.0, 1.0)
        pools = []
        for sd in range(3):
            p = gmm_pool(a.N, a.d, sd).to(dev)
            pools.append((p, p[:m].contiguous()))
        calls = [(pools[k % 3][0], pools[k % 3][1], a.n, kern) for k in range(a.batches)]
        seeds = [1] * a.batches

    def sequential():
        out = []
        for (pts, nys, n, k), sd in zip(calls, seeds):
            torch.manual_seed(sd)
            out.append(basq_amd.recombination(pts, nys, n, k, dev))
        torch.cuda.synchronize()
        return out

    ref = None
    if not a.pipelined_only:
        sequential()
        t0 = time.perf_counter()
        ref = sequential()
        seq = (time.perf_counter() - t0) / a.batches
        print(f"sequential           : {seq * 1e3:7.2f} ms/batch = {1 / seq:6.1f} batches/s", flush=True)
    for k in [int(v) for v in a.inflight.split(",") if v]:
        basq_amd.recombination_many(calls[:2 * k], dev, in_flight=k, seeds=seeds[:2 * k])      # warm the slots
        torch.cuda.synchronize()
        jobs = [Job(p, 0, p.shape[0], nys, n, kk, seed=sd) for (p, nys, n, kk), sd in zip(calls, seeds)]
        with _DEFAULT_POOL.lease(dev, k) as slots:
            t0 = time.perf_counter()
            res = RecombinationEngine(slots[0], LocalComm()).run_many(jobs, slots)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.batches
        lat = sorted(j.times["done"] - j.times["start"] for j in jobs)
        same = None if ref is None else all(torch.equal(i1, i2) and torch.equal(w1, w2) for (i1, w1), (i2, w2) in zip(ref, res))
        print(f"{k} in flight          : {dt * 1e3:7.2f} ms/batch = {1 / dt:6.1f} batches/s   latency median "
              f"{lat[len(lat) // 2] * 1e3:6.2f} ms, max {lat[-1] * 1e3:6.2f} ms   bit-identical to sequential: {same}", flush=True)


if __name__ == "__main__":
    main()
