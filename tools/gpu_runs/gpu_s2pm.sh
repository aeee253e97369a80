#!/bin/bash
# session-2: PMC passes on the shipped tall-skinny GEMM ([1e4,1e4] x [1e4,99], auto K split): HBM bytes, fp64-pipe busy cycles, clock
set -u
out=gpurun_out/s2pm; mkdir -p $out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$out/p_fetch -o p -- python3 $R/tools/prof_skinny.py --ksplit 6 > $R/$out/p_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$out/p_write -o p -- python3 $R/tools/prof_skinny.py --ksplit 6 > $R/$out/p_write.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/$out/p_pipe -o p -- python3 $R/tools/prof_skinny.py --ksplit 6 > $R/$out/p_pipe.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_INSTS_LDS --output-format csv -d $R/$out/p_inst -o p -- python3 $R/tools/prof_skinny.py --ksplit 6 > $R/$out/p_inst.log 2>&1
cd $R
python3 - <<'PY' > $out/summary.txt
import csv, glob, collections
agg = collections.defaultdict(list); dur = collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/s2pm/p_*/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        if 'skinny' not in r['Kernel_Name']: continue
        tr = 'A^T Q' if ', true>' in r['Kernel_Name'] else 'A Q'
        agg[(tr, r['Counter_Name'])].append(float(r['Counter_Value']))
        dur[tr].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for tr in ('A Q', 'A^T Q'):
    d = sorted(dur[tr]); print(tr, 'median duration under the profiler %.1f us' % (d[len(d) // 2] / 1e3))
    for (t, c), v in sorted(agg.items()):
        if t == tr: print('   %-28s %.5g' % (c, sorted(v)[len(v) // 2]))
PY
cat $out/summary.txt
