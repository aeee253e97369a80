#!/bin/bash
# round-2 GPU check #6: residue-class block sums + fixed panel Cholesky: tests, bench x2, per-config breakdown, profiles
set -u
out=gpurun_out/r02f; mkdir -p $out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "chol or trsm or blocksum or regroup or cluster" > $out/kernels.log 2>&1
echo "kernel tests rc=$?" >> $out/kernels.log; tail -4 $out/kernels.log
timeout 2400 python -m pytest tests -x -q -m gpu > $out/gpu_tests.log 2>&1
echo "gpu tests rc=$?" >> $out/gpu_tests.log; tail -6 $out/gpu_tests.log
timeout 900 python bench.py --breakdown > $out/bench1.json 2> $out/bench1.err; tail -3 $out/bench1.err | cut -c1-600; cut -c1-330 $out/bench1.json
timeout 900 python bench.py > $out/bench2.json 2> $out/bench2.err; cut -c1-330 $out/bench2.json
timeout 900 python tools/bench_configs.py > $out/configs.txt 2>&1; grep -v amdgpu.ids $out/configs.txt
python - > $out/build_old_apply.log 2>&1 <<'PY'
from basq_amd import _build
_build.build(force=True, verbose=False, defines={"BASQ_NS_APPLY16": 0}, out="/tmp/libbasq_oldapply.so")
PY
echo "== apply16 (default) 200 400" >> $out/reduction.txt; timeout 300 python tools/bench_reduction.py 200 400 2>&1 | grep -v amdgpu.ids >> $out/reduction.txt
echo "== 64-lane apply 200 400" >> $out/reduction.txt; BASQ_HIP_LIB=/tmp/libbasq_oldapply.so timeout 300 python tools/bench_reduction.py 200 400 2>&1 | grep -v amdgpu.ids >> $out/reduction.txt
cat $out/reduction.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_bench -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/prof_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_cfg4 -o cfg4 -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py --only cfg4_matern52_1e6_d32 > $GRAFT_REPO_ROOT/$out/prof_cfg4.log 2>&1
cd $GRAFT_REPO_ROOT
rm -f $out/prof_*/*trace.csv
for f in $(find $out -name "*kernel_stats.csv"); do echo $f; head -24 $f | cut -c1-150; done
