#!/bin/bash
# session-2 check k: every launch of two consecutive in-epoch rounds of a headline batch, with start offsets and gaps
set -u
out=gpurun_out/s2k; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -o b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - $(ls $out/prof/*kernel_trace.csv | head -1) <<'PY' > $out/seq.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
cm = [i for i, r in enumerate(rows) if 'col_mean' in r['Kernel_Name']]; start = cm[-2]; rows = rows[:cm[-1]]
pass
ns = [i for i, r in enumerate(rows) if 'bidiag' in r['Kernel_Name'] and i > start]
bs = [i for i, r in enumerate(rows) if 'blocksum_kernel' in r['Kernel_Name'] and i > start]
import os
win = os.environ.get('WIN', 'head')
a, b = (start, bs[0]) if win == 'head' else (len(rows) - 70, len(rows) - 1)
t0 = int(rows[a]['Start_Timestamp']); prev_end = t0
for r in rows[a:b + 1]:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%9.1f us  gap %6.1f  dur %7.1f us  %s' % ((st - t0) / 1e3, (st - prev_end) / 1e3, (en - st) / 1e3, r['Kernel_Name'][:60]))
    prev_end = en
PY
rm -f $out/prof/*trace.csv
cat $out/seq.txt
