#!/bin/bash
# round-2 GPU check #28: Cholesky panel kernel with the diagonal block factored by the row-owning waves only
set -u
out=gpurun_out/r02za; mkdir -p $out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "chol or range_finder or trsm" > $out/kernels.log 2>&1
echo "kernel tests rc=$?" >> $out/kernels.log; tail -3 $out/kernels.log | cut -c1-200
for shape in "100 200" "200 400"; do timeout 300 python tools/bench_reduction.py $shape 2>&1 | grep "chol" ; done
timeout 2400 python -m pytest tests -x -q -m gpu > $out/gpu_tests.log 2>&1
echo "gpu tests rc=$?" >> $out/gpu_tests.log; tail -4 $out/gpu_tests.log | cut -c1-300
timeout 900 python bench.py --no-cpu-baseline --steps 10 > $out/bench1.json 2> $out/bench1.err; cut -c80-140 $out/bench1.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_bench -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/prof_bench.log 2>&1
cd $GRAFT_REPO_ROOT
rm -f $out/prof_*/*trace.csv
grep "chol_factor" $out/prof_bench/bench_kernel_stats.csv | cut -c1-160
