"""The only configuration the reference publishes timings for: the tutorials'
SOBER-API batch selection -- n_cand = 20 000, n_nys = 500, batch 100, d = 10 (``SOBER/BASQ/_basq.py:19-36``) -- on one MI355X.

    python tools/bench_sober_tutorial.py [--reps 20]

Per case of ``oracle/make_golden_sober.TUTORIAL_CASES``: golden check (indices identical, weights), ms per batch of
``basq_amd.sober.recombination`` (one batch after the other), of the BASQ variant on the same inputs, and with two batches in
flight; beside them the imported reference's CPU seconds in the build container (recorded in the fixture) and the notebooks'
own "overhead" column (hardware unstated; it includes candidate sampling and gpytorch's autograd graph).
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import basq_amd                                                    # noqa: E402

basq_amd.configure_hw_queues()                                      # (before the first GPU call)
from basq_amd import sober                                         # noqa: E402
from oracle.make_golden_sober import TUTORIAL_CASES, tutorial_inputs   # noqa: E402
from tests.cases import build_product_kernel                      # noqa: E402

NOTEBOOK = {"tut01": "0.612 -> 1.680 s (n_obs 2 -> 902)", "tut02": "0.954 -> ~2.06 s", "tut03": "0.997 -> 2.886 s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    fx = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                                     "sober_tutorial.json")))
    for c, f in zip(TUTORIAL_CASES, fx):
        pts, nys = tutorial_inputs(c)
        pts, nys = pts.to(dev), nys.to(dev)
        kern = build_product_kernel(c)
        torch.manual_seed(1)
        idx, w = sober.recombination(pts, nys, c["n"], kern, dev, torch.float64)
        gw = torch.tensor(f["w"], dtype=torch.float64)
        same = idx.cpu().tolist() == f["idx"]
        rel = ((w.cpu() - gw).abs() / gw).max().item() if same else float("nan")

        def timed(fn):
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.reps):
                torch.manual_seed(1)
                fn()
            torch.cuda.synchronize()
            return 1e3 * (time.perf_counter() - t0) / a.reps

        ms_sober = timed(lambda: sober.recombination(pts, nys, c["n"], kern, dev, torch.float64))
        ms_basq = timed(lambda: basq_amd.recombination(pts, nys, c["n"], kern, dev))
        calls = [(pts, nys, c["n"], kern)] * 8
        basq_amd.recombination_many(calls, dev, in_flight=2, seeds=[1] * 8)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        basq_amd.recombination_many(calls, dev, in_flight=2, seeds=[1] * 8)
        torch.cuda.synchronize()
        ms_many = 1e3 * (time.perf_counter() - t0) / 8
        tag = c["name"].split("_")[1]
        print(f"{c['name']:32s} golden idx {same} rel {rel:.1e} | SOBER variant {ms_sober:7.2f} ms/batch | BASQ variant {ms_basq:7.2f} ms "
              f"(ratio {ms_sober / ms_basq:4.2f}) | BASQ, two in flight {ms_many:7.2f} ms | reference on this container's CPU "
              f"{f['reference_cpu_seconds_here']:.2f} s | notebook {NOTEBOOK.get(tag, '-')}", flush=True)


if __name__ == "__main__":
    main()
