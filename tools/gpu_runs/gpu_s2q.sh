#!/bin/bash
# session-2 check q: reflector rows in flight in nullspace_apply_kernel (4 / 8 / 12 / 16), kernel time by rocprofv3
set -u
out=gpurun_out/s2q; mkdir -p $out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in pf4 pf8 pf12 pf16; do
  BASQ_HIP_LIB=$R/tools/variants/$v.so rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/$v -o r -- python3 $R/tools/bench_reduction.py > $R/$out/$v.log 2>&1
  echo "$v $(grep nullspace_apply $R/$out/$v/r_kernel_stats.csv | cut -d, -f2-4) $(grep LAPACK $R/$out/$v.log)" >> $R/$out/ab.txt
  rm -f $R/$out/$v/*trace.csv
done
cat $R/$out/ab.txt
