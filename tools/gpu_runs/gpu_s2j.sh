#!/bin/bash
# session-2 check j: per-launch durations of the GEMM kernels inside one headline batch
set -u
out=gpurun_out/s2j; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -o b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - $(ls $out/prof/*kernel_trace.csv | head -1) <<'PY' > $out/seq.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last batch: from the last pack_points / col_mean launch on
start = max(i for i, r in enumerate(rows) if 'col_mean' in r['Kernel_Name'])
t0 = int(rows[start]['Start_Timestamp'])
for r in rows[start:]:
    n = r['Kernel_Name']
    if any(k in n for k in ('skinny', 'gemm_kernel', 'project', 'blocksum')):
        print('%9.1f us  +%8.1f us  grid %-10s %s' % ((int(r['Start_Timestamp']) - t0) / 1e3,
              (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r.get('Grid_Size', r.get('Grid_Size_X', '?')), n[:70]))
PY
rm -f $out/prof/*trace.csv
cat $out/seq.txt
