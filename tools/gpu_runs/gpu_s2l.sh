#!/bin/bash
# session-2 check l: one-round K split of the batched projection + test-matrix copy on its own stream: full suite, bench x2, configs
set -u
out=gpurun_out/s2l; mkdir -p $out
timeout -k 10 1500 python -m pytest tests -x -q -m gpu > $out/gpu_tests.log 2>&1
rc=$?; echo "gpu tests rc=$rc" >> $out/gpu_tests.log; tail -3 $out/gpu_tests.log | cut -c1-300
[ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" $out/gpu_tests.log | head -20 | cut -c1-250; exit 1; }
timeout -k 10 600 python bench.py --no-cpu-baseline --breakdown > $out/bench1.json 2> $out/bench1.err; cut -c1-330 $out/bench1.json
timeout -k 10 600 python bench.py --no-cpu-baseline > $out/bench2.json 2> $out/bench2.err; cut -c1-200 $out/bench2.json
timeout -k 10 900 python tools/bench_configs.py > $out/configs.txt 2>&1; grep -v amdgpu.ids $out/configs.txt | cut -c1-250
