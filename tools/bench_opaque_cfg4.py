"""BASELINE config 4 (Matern-5/2, N=1e6, d=32, n=200, m=1e4) through a BARE callable -- the reference's own ``kernel``
contract (``BASQ/_rchq.py:8,16``, tutorial 02) -- against its reference-generated golden, with the time per batch and the
achieved HBM rate of ``dense_blocksum_kernel`` (HBM-bound by design: it reads the 8-byte kernel value of every pair once).

    python tools/bench_opaque_cfg4.py [--reps 2]

The callable is a plain lambda over device tensor operations (the oracle's Matern closed form moved to the GPU); the
probe of ``CallableKernel`` decides between the reference's block-by-block calls and large chunks.
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import basq_amd                                                    # noqa: E402
from tests.cases import BY_NAME, build_oracle_kernel, build_pool, load_golden   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--name", default="cfg4_matern52_1e6_d32")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    c, fx = BY_NAME[a.name], load_golden(a.name)
    pts, nys = build_pool(c)
    pts, nys = pts.to(dev), nys.to(dev)
    ko, _ = build_oracle_kernel(c)
    kern = lambda x, y: ko(x, y)                                   # noqa: E731  a bare lambda, nothing else
    from basq_amd._ops import HipOps
    from basq_amd.kernels import CallableKernel

    exact = CallableKernel(kern).resolve_mode(HipOps(dev), nys, 2 * c["n"])
    print(f"probe: block_exact = {exact}")
    tr = basq_amd.EngineTrace(host_sync=True)
    torch.manual_seed(c["torch_seed"])
    t0 = time.perf_counter()
    idx, w = basq_amd.recombination(pts, nys, c["n"], kern, dev, trace=tr)
    torch.cuda.synchronize()
    first = time.perf_counter() - t0
    gw = torch.tensor(fx["w"], dtype=torch.float64)
    same = idx.cpu().tolist() == fx["idx"]
    rel = ((w.cpu() - gw).abs() / gw).max().item()
    kept_ok = [r["kept"] for r in tr.rounds] == [r["kept"] for r in fx["rounds"]]
    print(f"golden {a.name}: indices identical = {same}, per-round kept sets identical = {kept_ok}, max rel weight error = {rel:.2e} "
          f"({len(tr.rounds)} rounds); first (traced) batch {first:.2f} s; phases: "
          + "  ".join(f"{k} {v:.3f}" for k, v in tr.timers.items() if k in ("basis", "blocksum", "project", "nullspace", "eliminate", "compact")))
    ts = []
    for _ in range(a.reps):
        torch.manual_seed(c["torch_seed"])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        basq_amd.recombination(pts, nys, c["n"], kern, dev)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    pairs = float(c["m"]) * sum(r["R"] for r in tr.rounds)
    print(f"untraced: {min(ts) * 1e3:.1f} ms/batch (best of {a.reps}); {pairs:.3e} kernel values per batch -> the dense block sums read "
          f"{8 * pairs / 1e9:.1f} GB per batch; whole batch = {8 * pairs / min(ts) / 1e9:.0f} GB/s of kernel values consumed "
          f"(the callable's own tensor operations write and re-read each value several times on top)")


if __name__ == "__main__":
    main()
