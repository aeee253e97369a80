"""Achieved HBM rate of ``basq_dense_blocksum_f64`` (the block sums of an opaque callable's kernel values: 8 bytes per pair,
read once) on chunks of the size the engine hands it (1 GB of kernel values).

    BASQ_DBS_NS=<slices> python tools/bench_dense_blocksum.py [--m 10000] [--S 400] [--reps 20]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from basq_amd._ops import HipOps                                   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=10_000)
    ap.add_argument("--S", type=int, default=400)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--fresh", action="store_true", help="as inside a batch: every chunk is WRITTEN by an element-wise kernel right "
                    "before its block sums (three buffers in rotation), only the block sums are timed")
    a = ap.parse_args()
    ops = HipOps(torch.device("cuda", 0))
    nc = ((1 << 30) // (8 * a.m) // a.S) * a.S
    Cm = torch.rand(a.m, nc, dtype=torch.float64, device=ops.device)
    mu = torch.rand(nc, dtype=torch.float64, device=ops.device)
    E, T = ops.zeros(a.m, a.S), ops.zeros(a.S)
    n_full = nc * 4
    for _ in range(3):
        ops.dense_blocksum(Cm, mu, 0, n_full, a.S, 1.0, E, tot=T)
    if a.fresh:
        bufs = [Cm, Cm.clone(), Cm.clone()]
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.reps)]
        for k in range(a.reps):
            Ck = bufs[k % 3]
            Ck.mul_(1.0000001)
            ev[k][0].record()
            ops.dense_blocksum(Ck, mu, 0, n_full, a.S, 1.0, E, tot=T)
            ev[k][1].record()
        torch.cuda.synchronize()
        ms = sorted(x.elapsed_time(y) for x, y in ev)[a.reps // 2]
    else:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            ops.dense_blocksum(Cm, mu, 0, n_full, a.S, 1.0, E, tot=T)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
    gb = 8.0 * a.m * nc / 1e9
    print(f"{'fresh ' if a.fresh else 'repeat'} NT={os.environ.get('BASQ_DBS_NT', '0')} BASQ_DBS_NS={os.environ.get('BASQ_DBS_NS', 'auto'):>4s}  m={a.m} S={a.S} nc={nc}: {ms * 1e3:7.1f} us per chunk of {gb:.3f} GB "
          f"-> {gb / ms:6.2f} TB/s of kernel values ({(gb + 16e-9 * a.m * a.S) / ms:5.2f} TB/s with the read-modify-write of the sums)", flush=True)


if __name__ == "__main__":
    main()
