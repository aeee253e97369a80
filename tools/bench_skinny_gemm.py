"""Timing of the range finder's products: basq_skinny_gemm_f64 (K-split sweep) against the library paths.

    python tools/bench_skinny_gemm.py [--m 10000] [--q 99]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from basq_amd._ops import HipOps          # noqa: E402


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=10000)
    ap.add_argument("--q", type=int, default=99)
    a = ap.parse_args()
    ops = HipOps("cuda:0")
    g = torch.Generator().manual_seed(0)
    A = torch.randn(a.m, a.m, generator=g, dtype=torch.float64).to("cuda:0")
    Q = torch.randn(a.m, a.q, generator=g, dtype=torch.float64).to("cuda:0")
    flops = 2.0 * a.m * a.m * a.q
    print(f"[{a.m} x {a.m}] @ [{a.m} x {a.q}]  ({flops / 1e9:.1f} GFLOP; at 47 TF/s: {flops / 47e12 * 1e6:.0f} us)")
    t = timed(lambda: torch.matmul(A, Q))
    print(f"  torch.matmul A @ Q              {t:8.1f} us  {flops / t / 1e6:6.1f} TF/s")
    t = timed(lambda: torch.matmul(A.t(), Q))
    print(f"  torch.matmul A^T @ Q            {t:8.1f} us  {flops / t / 1e6:6.1f} TF/s")
    Ab = A.unflatten(1, (8, a.m // 8)).permute(1, 0, 2)
    t = timed(lambda: torch.bmm(Ab, Q.unflatten(0, (8, a.m // 8))).sum(0))
    print(f"  bmm split-K 8 (round-1 path)    {t:8.1f} us  {flops / t / 1e6:6.1f} TF/s")
    for trans in (False, True):
        auto = ops._skinny_ksplit(a.m, a.m, a.q)
        for nz in sorted({1, 2, 3, 4, 6, 8, 13, 16, 26, auto}):
            t = timed(lambda: ops.skinny_gemm(A, Q, trans=trans, ksplit=nz))
            print(f"  skinny_gemm trans={int(trans)} ksplit={nz:3d}{' (auto)' if nz == auto else '       '} {t:8.1f} us  {flops / t / 1e6:6.1f} TF/s")
    X = Q
    fl2 = 2.0 * a.m * a.q * a.q
    t = timed(lambda: torch.matmul(X.t(), X))
    print(f"  X^T X torch.matmul              {t:8.1f} us")
    for nz in sorted({16, 32, 64, 125, ops._skinny_ksplit(a.q, a.m, a.q)}):
        t = timed(lambda: ops.skinny_gemm(X, X, trans=True, ksplit=nz))
        print(f"  X^T X skinny ksplit={nz:3d}         {t:8.1f} us  {fl2 / t / 1e6:6.1f} TF/s")


if __name__ == "__main__":
    main()
