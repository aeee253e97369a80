#!/bin/bash
# session-2 check n: four reflectors per reduction round in nullspace_apply_kernel: tests, reduction timing, bench
set -u
out=gpurun_out/s2n; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu > $out/kernel_tests.log 2>&1 || { tail -30 $out/kernel_tests.log | cut -c1-200; exit 1; }
tail -1 $out/kernel_tests.log
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_sober.py -x -q -m gpu > $out/parity.log 2>&1 || { tail -30 $out/parity.log | cut -c1-200; exit 1; }
tail -1 $out/parity.log
timeout -k 10 300 python tools/bench_reduction.py > $out/reduction.txt 2>&1; grep -v amdgpu.ids $out/reduction.txt | cut -c1-200
timeout -k 10 600 python bench.py --no-cpu-baseline > $out/bench1.json 2> $out/bench1.err; cut -c1-250 $out/bench1.json
