"""Launch-by-launch timeline of one batch from a rocprofv3 kernel trace (``--kernel-trace --output-format csv``).

    python tools/trace_round.py <..._kernel_trace.csv> [--batch -2] [--from-kernel bidiag --count 3]

Prints every kernel of the chosen batch (batches start at ``col_mean_kernel``) between the k-th and (k+count)-th launch of
``--from-kernel``: start offset, duration, and the gap since the previous kernel ended -- what one round of the
divide-and-conquer loop costs beyond its two reductions.
"""
import argparse
import csv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--batch", type=int, default=-2)
    ap.add_argument("--from-kernel", default="bidiag")
    ap.add_argument("--first", type=int, default=1, help="start at this occurrence of --from-kernel (0-based)")
    ap.add_argument("--count", type=int, default=2)
    a = ap.parse_args()
    rows = []
    with open(a.csv) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:58]))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if "col_mean" in r[2]]
    b = starts[a.batch]
    e = starts[a.batch + 1] if a.batch + 1 < len(starts) and a.batch != -1 else len(rows)
    batch = rows[b:e]
    marks = [i for i, r in enumerate(batch) if a.from_kernel in r[2]]
    lo, hi = marks[a.first], marks[min(a.first + a.count, len(marks) - 1)]
    t0 = batch[lo][0]
    prev_end = batch[lo - 1][1] if lo > 0 else t0
    tot_k = tot_g = 0
    for s, en, n in batch[lo:hi]:
        gap = s - prev_end
        print(f"  +{(s - t0) / 1e3:9.1f} us  {n:58s} {(en - s) / 1e3:8.1f} us   gap {gap / 1e3:6.1f} us")
        tot_k += en - s
        tot_g += max(gap, 0)
        prev_end = max(prev_end, en)
    print(f"{a.count} round(s): span {(batch[hi][0] - t0) / 1e3:.1f} us, kernels {tot_k / 1e3:.1f} us, gaps {tot_g / 1e3:.1f} us")
    whole = batch[-1][1] - batch[0][0]
    busy = sum(en - s for s, en, _ in batch)
    print(f"whole batch: span {whole / 1e6:.3f} ms, kernel time {busy / 1e6:.3f} ms, {len(batch)} launches")


if __name__ == "__main__":
    main()
