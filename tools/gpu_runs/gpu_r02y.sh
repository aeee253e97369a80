#!/bin/bash
# round-2 GPU check #25: 6 instead of 16 K slices for the range finder's products; full suite + bench + A/B vs library
set -u
out=gpurun_out/r02y; mkdir -p $out
timeout 2400 python -m pytest tests -x -q -m gpu > $out/gpu_tests.log 2>&1
echo "gpu tests rc=$?" >> $out/gpu_tests.log; tail -4 $out/gpu_tests.log | cut -c1-300
timeout 600 python tools/ab_engine.py OWN_RANGE_GEMM 0 1 --reps 8 > $out/ab_own_gemm.txt 2>&1; grep -v amdgpu.ids $out/ab_own_gemm.txt | tail -2
timeout 900 python bench.py --no-cpu-baseline --steps 10 > $out/bench1.json 2> $out/bench1.err; cut -c80-140 $out/bench1.json
timeout 1200 python tools/bench_configs.py --only cfg4_matern52_1e6_d32 2>&1 | grep -v amdgpu.ids | cut -c1-140
