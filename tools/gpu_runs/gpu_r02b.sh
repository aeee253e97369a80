#!/bin/bash
# round-2 GPU check of the cluster reduction kernels: parity first, then timing (new vs round-1 kernels)
set -u
out=gpurun_out/r02b; mkdir -p $out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "nullspace or car_eliminate or cluster or fuzz" > $out/kernels.log 2>&1
echo "kernel tests rc=$?" >> $out/kernels.log
tail -5 $out/kernels.log
python - > $out/build_old.log 2>&1 <<'PY'
from basq_amd import _build
_build.build(force=True, verbose=False, defines={"BASQ_NS_CLUSTER": 0, "BASQ_CAR_CLUSTER": 0, "BASQ_NS_APPLY16": 0}, out="/tmp/libbasq_old.so")
PY
for shape in "100 200" "200 400" "50 100" "31 62"; do
  echo "== new $shape" >> $out/reduction.txt; timeout 300 python tools/bench_reduction.py $shape >> $out/reduction.txt 2>&1
  echo "== old $shape" >> $out/reduction.txt; BASQ_HIP_LIB=/tmp/libbasq_old.so timeout 300 python tools/bench_reduction.py $shape >> $out/reduction.txt 2>&1
done
cat $out/reduction.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/prof_red -o red -- python3 $GRAFT_REPO_ROOT/tools/bench_reduction.py 100 200 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/prof_red4 -o red -- python3 $GRAFT_REPO_ROOT/tools/bench_reduction.py 200 400 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find $out -name "*kernel_stats.csv" | head; for f in $(find $out -name "*kernel_stats.csv"); do head -8 $f | cut -c1-200; done
