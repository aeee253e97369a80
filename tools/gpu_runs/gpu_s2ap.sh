#!/bin/bash
# session-2: reflector application at 200 x 400: 16 lanes per null vector (4 rows in flight) against 64 lanes (12 rows in flight)
set -u
out=gpurun_out/s2ap; mkdir -p $out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in ap16 ap64; do
  BASQ_HIP_LIB=$R/tools/variants/$v.so rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/$v -o r -- python3 $R/tools/bench_reduction.py 200 400 --reps 20 > $R/$out/$v.log 2>&1
  echo "$v $(grep nullspace_apply $R/$out/$v/r_kernel_stats.csv | sed 's/.*)",//' | cut -c1-70) | $(grep LAPACK $R/$out/$v.log)" >> $R/$out/ab.txt
  rm -f $R/$out/$v/*trace.csv
done
cat $R/$out/ab.txt
