"""Host-side timeline of consecutive synchronous calls: where the time between the last kernel of one batch and the first kernel
of the next goes (entry -> first launch, every wait of the batch, last wake-up -> return, return -> next entry).

    python tools/host_timeline.py [--batches 12] [--case cfg2_rbf_1e5]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import basq_amd                                   # noqa: E402
import basq_amd._basis as basis                   # noqa: E402
from basq_amd._ops import HipOps                  # noqa: E402
from basq_amd.pools import gmm_pool               # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=12)
    ap.add_argument("--N", type=int, default=1_000_000)
    ap.add_argument("--case", default="")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    if a.case:
        from tests.cases import BY_NAME, build_pool
        from basq_amd.pools import kernel_for_case

        c = BY_NAME[a.case]
        pts, nys = build_pool(c)
        pool, nys, n, kern, seed = pts.to(dev), nys.to(dev), c["n"], kernel_for_case(c), c["torch_seed"]
    else:
        pool = gmm_pool(a.N, 10, 0).to(dev)
        nys, n, kern, seed = pool[:a.N // 100].contiguous(), 100, basq_amd.kernels.StationaryKernel("rbf", 2.0, 1.0), 1
    marks = []
    real_wait, real_col_mean = basis.wait_for, HipOps.col_mean

    def wait_for(ev):
        t0 = time.perf_counter()
        real_wait(ev)
        marks.append(("wait", t0, time.perf_counter()))

    def col_mean(self, X):
        marks.append(("first_launch", time.perf_counter(), 0.0))
        return real_col_mean(self, X)

    basis.wait_for = wait_for
    import basq_amd._engine as eng_mod
    for mod in (basis, eng_mod):
        if hasattr(mod, "wait_for"):
            mod.wait_for = wait_for
    HipOps.col_mean = col_mean
    rows = []
    prev_ret = None
    for k in range(a.batches + 2):
        del marks[:]
        t_in = time.perf_counter()
        torch.manual_seed(seed)
        t_seed = time.perf_counter()
        basq_amd.recombination(pool, nys, n, kern, dev)
        t_out = time.perf_counter()
        first = next(t for name, t, _ in marks if name == "first_launch")
        waits = [(t0, t1) for name, t0, t1 in marks if name == "wait"]
        rows.append(dict(gap_from_prev=(t_in - prev_ret) if prev_ret else 0.0, seed=t_seed - t_in, entry_to_first_launch=first - t_seed,
                         n_waits=len(waits), waits_ms=[round(1e3 * (b - a_), 3) for a_, b in waits],
                         last_wake_to_return=t_out - waits[-1][1], total=t_out - t_in))
        prev_ret = t_out
    rows = rows[2:]
    us = lambda key: sorted(1e6 * r[key] for r in rows)[len(rows) // 2]          # noqa: E731
    print(f"median over {len(rows)} batches: caller's loop between calls {us('gap_from_prev'):.0f} us, torch.manual_seed {us('seed'):.0f} us, "
          f"entry -> first launch {us('entry_to_first_launch'):.0f} us, last wake-up -> return {us('last_wake_to_return'):.0f} us, "
          f"call {us('total') / 1e3:.3f} ms, {rows[0]['n_waits']} waits")
    print("waits of one batch (ms):", rows[len(rows) // 2]["waits_ms"])


if __name__ == "__main__":
    main()
