"""Timing of the block-sum kernel on one GPU (HIP events on the launch stream).

    python tools/bench_blocksum.py [--R 1000000] [--m 10000] [--d 10] [--n 100] [--family rbf]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from basq_amd._ops import HipOps                      # noqa: E402
from basq_amd._partition import RoundGeometry, choose_chunks, local_blocks   # noqa: E402
from basq_amd.kernels import StationaryKernel         # noqa: E402
from basq_amd.pools import gmm_pool                   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--R", type=int, default=1_000_000)
    ap.add_argument("--m", type=int, default=10_000)
    ap.add_argument("--d", type=int, default=10)
    ap.add_argument("--n", type=int, default=100)
    ap.add_argument("--family", default="rbf")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--chunks", type=int, default=0)
    ap.add_argument("--classes", type=int, default=16, help="residue classes of the block index (what round 1 of an "
                    "epoch launches; 0 = contiguous chunks)")
    args = ap.parse_args()
    ops = HipOps("cuda:0")
    S = 2 * args.n
    spec = StationaryKernel(args.family, 2.0).spec(args.d)
    pts = ops.to_device(gmm_pool(args.R, args.d, 0))
    nys = pts[: args.m].contiguous()
    c = ops.col_mean(nys)
    A = ops.pack(spec, nys, c, 0, pad_rows_to=64)
    B = ops.pack(spec, pts, c, 1)
    mu, _ = ops.init_state(args.R, 0, args.R)
    geo = RoundGeometry.of(args.R, S)
    pairs = float(args.R) * args.m
    out = {}
    impls = ("mfma",)
    for impl in impls:
        cm = args.classes
        if cm:
            nch, Rr = cm, (geo.nb // cm) * cm * S          # the regular region: a multiple of cm full blocks
        else:
            nch, Rr = args.chunks or choose_chunks(local_blocks(0, args.R, geo), args.m, S, ops.kp(args.d) // 4), args.R
        pairs = float(Rr) * args.m
        X, t = ops.blocksum(spec, A, args.m, B, mu, None, Rr, 0, geo.n_full, S, nch, class_mod=cm)   # warm-up
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            X, t = ops.blocksum(spec, A, args.m, B, mu, None, Rr, 0, geo.n_full, S, nch, class_mod=cm)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.reps
        out[impl] = X.sum(0)
        flops = pairs * (3 * args.d + 3)
        print(f"{impl:5s} chunks={nch} {ms:9.3f} ms  {pairs / ms / 1e6:8.2f} Gpair/s  {flops / ms / 1e9:7.2f} TFLOP/s(3d+3)")


if __name__ == "__main__":
    main()
