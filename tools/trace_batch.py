"""Whole-batch timeline summary from a rocprofv3 kernel trace: per batch (batches start at ``col_mean_kernel``) the span, the
kernel time, the launches and the gaps; per kernel name the time and the gap that PRECEDES its launches (last full batch).

    python tools/trace_batch.py <..._kernel_trace.csv> [--batch -2] [--list]
"""
import argparse
import csv
from collections import defaultdict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--batch", type=int, default=-2)
    ap.add_argument("--list", action="store_true", help="every launch of the batch")
    a = ap.parse_args()
    rows = []
    with open(a.csv) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if "col_mean" in r[2]]
    spans = []
    for k in range(len(starts) - 1):
        b = rows[starts[k]:starts[k + 1]]
        spans.append((b[-1][1] - b[0][0], sum(e - s for s, e, _ in b), len(b), rows[starts[k + 1]][0] - b[0][0]))
    for k, (sp, busy, n, period) in enumerate(spans):
        print(f"batch {k:3d}: first launch -> last end {sp / 1e6:7.3f} ms, kernels {busy / 1e6:7.3f} ms, gaps {(sp - busy) / 1e6:6.3f} ms, "
              f"{n} launches, period to the next batch {period / 1e6:7.3f} ms")
    k = a.batch % (len(starts) - 1)
    b = rows[starts[k]:starts[k + 1]]
    t_k, t_g, cnt = defaultdict(int), defaultdict(int), defaultdict(int)
    prev = b[0][0]
    for s, e, n in b:
        t_k[n] += e - s
        t_g[n] += max(s - prev, 0)
        cnt[n] += 1
        if a.list:
            print(f"  +{(s - b[0][0]) / 1e3:9.1f} us  {n:60s} {(e - s) / 1e3:8.1f} us   gap {max(s - prev, 0) / 1e3:6.1f} us")
        prev = max(prev, e)
    print(f"batch {k}: per kernel (time, gap BEFORE its launches)")
    for n in sorted(t_k, key=lambda x: -(t_k[x] + t_g[x])):
        print(f"  {n:60s} x{cnt[n]:4d}  {t_k[n] / 1e3:9.1f} us  + gaps {t_g[n] / 1e3:8.1f} us")


if __name__ == "__main__":
    main()
