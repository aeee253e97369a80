#!/bin/bash
# round-2 GPU check #5: microbench, kernel + full GPU test suite, reductions timing, bench, per-config breakdown
set -u
out=gpurun_out/r02e; mkdir -p $out
./tools/microbench > $out/microbench.txt 2>&1; grep "waves/SIMD=4" $out/microbench.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "chol or trsm or nullspace or car_eliminate or cluster or fuzz or blocksum" > $out/kernels.log 2>&1
echo "kernel tests rc=$?" >> $out/kernels.log; tail -4 $out/kernels.log
for shape in "100 200" "200 400"; do echo "== $shape" >> $out/reduction.txt; timeout 300 python tools/bench_reduction.py $shape 2>&1 | grep -v amdgpu.ids >> $out/reduction.txt; done
cat $out/reduction.txt
timeout 2400 python -m pytest tests -x -q -m gpu > $out/gpu_tests.log 2>&1
echo "gpu tests rc=$?" >> $out/gpu_tests.log; tail -6 $out/gpu_tests.log
timeout 900 python bench.py --breakdown > $out/bench1.json 2> $out/bench1.err; tail -3 $out/bench1.err; cut -c1-400 $out/bench1.json
timeout 900 python tools/bench_configs.py > $out/configs.txt 2>&1; grep -v amdgpu.ids $out/configs.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_bench -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/prof_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_cfg4 -o cfg4 -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py --only cfg4_matern52_1e6_d32 > $GRAFT_REPO_ROOT/$out/prof_cfg4.log 2>&1
cd $GRAFT_REPO_ROOT
for f in $(find $out -name "*kernel_stats.csv"); do echo $f; head -16 $f | cut -c1-150; done
