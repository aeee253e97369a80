"""A handful of basq_skinny_gemm_f64 launches for rocprofv3 (kernel trace or PMC passes).

    python tools/prof_skinny.py [--m 10000] [--q 99] [--ksplit 16] [--reps 5]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from basq_amd._ops import HipOps          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=10000)
    ap.add_argument("--q", type=int, default=99)
    ap.add_argument("--ksplit", type=int, default=16)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    ops = HipOps("cuda:0")
    g = torch.Generator().manual_seed(0)
    A = torch.randn(a.m, a.m, generator=g, dtype=torch.float64).to("cuda:0")
    Q = torch.randn(a.m, a.q, generator=g, dtype=torch.float64).to("cuda:0")
    for trans in (False, True):
        for _ in range(a.reps):
            ops.skinny_gemm(A, Q, trans=trans, ksplit=a.ksplit)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
