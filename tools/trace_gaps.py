"""GPU idle time between kernels, from a rocprofv3 kernel trace (``--kernel-trace --output-format csv``).

    python tools/trace_gaps.py <..._kernel_trace.csv> [--skip-first-ms 0]

Prints the busy / idle split of the traced span and, per kernel name, the idle time that FOLLOWS its launches (the gap
until the next kernel starts): which hand-overs leave the GPU waiting for the host.
"""
import argparse
import csv
from collections import defaultdict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--tail-fraction", type=float, default=0.6, help="analyse only the last fraction of the trace (steady state)")
    a = ap.parse_args()
    rows = []
    with open(a.csv) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    rows = rows[int(len(rows) * (1.0 - a.tail_fraction)):]
    span = rows[-1][1] - rows[0][0]
    busy, idle_after, count = 0, defaultdict(int), defaultdict(int)
    end = rows[0][0]
    prev = None
    for s, e, name in rows:
        if s > end:
            if prev is not None:
                idle_after[prev] += s - end
        if e > end:
            busy += e - max(s, end)
            end = e
        prev = name.split("(")[0][:70]
        count[prev] += 1
    print(f"kernels {len(rows)}  span {span / 1e6:.2f} ms  busy {busy / 1e6:.2f} ms  idle {(span - busy) / 1e6:.2f} ms "
          f"({100.0 * (span - busy) / span:.1f} %)")
    for name, ns in sorted(idle_after.items(), key=lambda kv: -kv[1])[:14]:
        print(f"  idle after {name:70s} {ns / 1e6:8.3f} ms over {count[name]:5d} launches ({ns / 1e3 / max(count[name], 1):7.1f} us each)")


if __name__ == "__main__":
    main()
