#!/bin/bash
# session-2 check a: block sums on v_mfma_f64_4x4x4_4b -- kernel tests, kernel timing, headline bench
set -u
out=gpurun_out/s2a; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu > $out/kernel_tests.log 2>&1
echo "kernel tests rc=$?" >> $out/kernel_tests.log; tail -3 $out/kernel_tests.log | cut -c1-300
timeout -k 10 300 python tools/bench_blocksum.py > $out/bench_blocksum.txt 2>&1; grep -v amdgpu.ids $out/bench_blocksum.txt | cut -c1-200
timeout -k 10 300 python tools/bench_blocksum.py --d 32 --n 200 --family matern52 > $out/bench_blocksum_cfg4.txt 2>&1; grep -v amdgpu.ids $out/bench_blocksum_cfg4.txt | cut -c1-200
timeout -k 10 600 python bench.py --no-cpu-baseline --breakdown > $out/bench1.json 2> $out/bench1.err; cut -c1-400 $out/bench1.json
