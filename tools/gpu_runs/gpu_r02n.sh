#!/bin/bash
# round-2 GPU check #14: own tall-skinny GEMM for the range finder (basq_skinny_gemm_f64) -- parity, bench, stats
set -u
out=gpurun_out/r02n; mkdir -p $out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "skinny or range_finder" > $out/kernels.log 2>&1
echo "kernel tests rc=$?" >> $out/kernels.log; tail -3 $out/kernels.log
timeout 900 python bench.py --breakdown > $out/bench1.json 2> $out/bench1.err; cut -c1-330 $out/bench1.json
timeout 900 python bench.py --no-cpu-baseline > $out/bench2.json 2> $out/bench2.err; cut -c1-200 $out/bench2.json
timeout 2400 python -m pytest tests -x -q -m gpu > $out/gpu_tests.log 2>&1
echo "gpu tests rc=$?" >> $out/gpu_tests.log; tail -5 $out/gpu_tests.log | cut -c1-300
timeout 1200 python tools/bench_configs.py > $out/configs.txt 2>&1; grep -v amdgpu.ids $out/configs.txt | cut -c1-330
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_bench -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/prof_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_cfg4 -o cfg4 -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py --only cfg4_matern52_1e6_d32 > $GRAFT_REPO_ROOT/$out/prof_cfg4.log 2>&1
cd $GRAFT_REPO_ROOT
rm -f $out/prof_*/*trace.csv
head -14 $out/prof_bench/bench_kernel_stats.csv | cut -c1-150; head -9 $out/prof_cfg4/cfg4_kernel_stats.csv | cut -c1-150
