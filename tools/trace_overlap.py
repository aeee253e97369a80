"""Do the single-work-group reductions of one batch run INSIDE the wide kernels of another?  From a rocprofv3 kernel trace
(``--kernel-trace --output-format csv``).

    python tools/trace_overlap.py <..._kernel_trace.csv> [--tail-fraction 0.6]

Prints, for the steady-state part of the trace: the busy / idle split of the GPU, the time during which >= 2 kernels were
in flight, and per "chain" kernel (bidiagonalisation, null-space apply, elimination, Cholesky, triangular solve) the share
of its run time during which a WIDE kernel (block sums, tall-skinny GEMM) of another queue was running -- plus the queue ids
seen, so that one can check the two batches really sat on different hardware queues.
"""
import argparse
import csv
from collections import defaultdict

CHAIN = ("bidiag", "nullspace_apply", "car_eliminate", "chol_", "trsm_rows")
WIDE = ("blocksum", "skinny_gemm", "gram_kernel", "dense_blocksum")


def kind(name):
    if any(k in name for k in CHAIN):
        return "chain"
    if any(k in name for k in WIDE):
        return "wide"
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--tail-fraction", type=float, default=0.6)
    a = ap.parse_args()
    rows = []
    with open(a.csv) as f:
        for r in csv.DictReader(f):
            q = r.get("Queue_Id") or r.get("Stream_Id") or "0"
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60], q))
    rows.sort()
    rows = rows[int(len(rows) * (1.0 - a.tail_fraction)):]
    t_lo, t_hi = rows[0][0], max(r[1] for r in rows)
    # sweep: number of kernels in flight
    ev = []
    for s, e, _, _ in rows:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    depth, last, busy, multi = 0, t_lo, 0, 0
    for t, dlt in ev:
        if depth >= 1:
            busy += t - last
        if depth >= 2:
            multi += t - last
        depth += dlt
        last = t
    span = t_hi - t_lo
    queues = sorted({r[3] for r in rows})
    print(f"kernels {len(rows)}  span {span / 1e6:.2f} ms  busy {busy / 1e6:.2f} ms ({100 * busy / span:.1f} %)  "
          f">=2 kernels in flight {multi / 1e6:.2f} ms ({100 * multi / span:.1f} %)  queues {queues}")
    wide = [(s, e, q) for s, e, n, q in rows if kind(n) == "wide"]
    tot, cov, cnt = defaultdict(int), defaultdict(int), defaultdict(int)
    wsum = defaultdict(int)
    for s, e, n, q in rows:
        wsum[n] += e - s
        if kind(n) != "chain":
            continue
        tot[n] += e - s
        cnt[n] += 1
        # union of the other queues' wide kernels clipped to [s, e)
        segs = sorted((max(s, ws), min(e, we)) for ws, we, wq in wide if wq != q and ws < e and we > s)
        c, end = 0, s
        for ls, le in segs:
            if le > end:
                c += le - max(ls, end)
                end = le
        cov[n] += c
    print("chain kernels: share of their run time with a wide kernel of ANOTHER queue in flight")
    for n in sorted(tot, key=lambda k: -tot[k]):
        print(f"  {n:60s} {cnt[n]:5d} launches {tot[n] / 1e6:8.3f} ms  overlapped {100.0 * cov[n] / max(tot[n], 1):5.1f} %  "
              f"avg {tot[n] / 1e3 / cnt[n]:7.1f} us")
    print("kernel time by name (ms):")
    for n, v in sorted(wsum.items(), key=lambda kv: -kv[1])[:12]:
        print(f"  {n:60s} {v / 1e6:9.3f}")


if __name__ == "__main__":
    main()
