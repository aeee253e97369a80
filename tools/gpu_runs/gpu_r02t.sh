#!/bin/bash
# round-2 GPU check #20: panel form of the LDS elimination -- bit-exactness, timing vs the step-by-step kernel, bench
set -u
out=gpurun_out/r02t; mkdir -p $out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "car_eliminate or fuzz or cluster" > $out/kernels.log 2>&1
echo "kernel tests rc=$?" >> $out/kernels.log; tail -12 $out/kernels.log | cut -c1-200
for shape in "100 200" "50 100" "30 200"; do echo "== $shape (panel)" >> $out/reduction.txt; timeout 300 python tools/bench_reduction.py $shape 2>&1 | grep "car_eliminate" >> $out/reduction.txt; echo "== $shape (step by step)" >> $out/reduction.txt; BASQ_CAR_PANEL=0 timeout 300 python tools/bench_reduction.py $shape 2>&1 | grep "car_eliminate" >> $out/reduction.txt; done
cat $out/reduction.txt
timeout 900 python bench.py --no-cpu-baseline --steps 10 > $out/bench1.json 2> $out/bench1.err; cut -c80-140 $out/bench1.json
BASQ_CAR_PANEL=0 timeout 900 python bench.py --no-cpu-baseline --steps 10 > $out/bench0.json 2> $out/bench0.err; cut -c80-140 $out/bench0.json
timeout 2400 python -m pytest tests -x -q -m gpu > $out/gpu_tests.log 2>&1
echo "gpu tests rc=$?" >> $out/gpu_tests.log; tail -5 $out/gpu_tests.log | cut -c1-300
