#!/bin/bash
# session-2 check i: class-partial projections on the batched tall-skinny kernel -- kernel + parity tests, bench, configs, kernel stats
set -u
out=gpurun_out/s2i; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu > $out/kernel_tests.log 2>&1 || { tail -30 $out/kernel_tests.log | cut -c1-200; exit 1; }
tail -1 $out/kernel_tests.log
timeout -k 10 1500 python -m pytest tests -x -q -m gpu --deselect tests/test_kernels_gpu.py > $out/gpu_tests.log 2>&1
rc=$?; echo "gpu tests rc=$rc" >> $out/gpu_tests.log; tail -3 $out/gpu_tests.log | cut -c1-300
[ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" $out/gpu_tests.log | head -20 | cut -c1-250; exit 1; }
timeout -k 10 600 python bench.py --no-cpu-baseline --breakdown > $out/bench1.json 2> $out/bench1.err; cut -c1-330 $out/bench1.json
timeout -k 10 900 python tools/bench_configs.py > $out/configs.txt 2>&1; grep -v amdgpu.ids $out/configs.txt | cut -c1-250
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_bench -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/prof_bench.log 2>&1
cd $GRAFT_REPO_ROOT
rm -f $out/prof_*/*trace.csv
head -16 $out/prof_bench/bench_kernel_stats.csv | cut -c1-60,120-220
