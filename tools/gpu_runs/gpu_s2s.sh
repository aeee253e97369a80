#!/bin/bash
# session-2 check s: tall-skinny GEMM with exact column-group counts (6 tiles + 1 group at q = 99, 12 + 2 at q = 199)
set -u
out=gpurun_out/s2s; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "skinny or project" > $out/tests_skinny.log 2>&1 || { tail -30 $out/tests_skinny.log | cut -c1-200; exit 1; }
tail -1 $out/tests_skinny.log
timeout -k 10 200 python tools/bench_skinny_gemm.py 2>&1 | grep -v amdgpu.ids | grep "skinny\|X^T" | grep "auto\|= 16\|= 62\|= 32\|=125" > $out/bench.txt; cat $out/bench.txt
timeout -k 10 200 python tools/bench_skinny_gemm.py --q 199 2>&1 | grep -v amdgpu.ids | grep "auto\|=  8" > $out/bench199.txt; cat $out/bench199.txt
timeout -k 10 1500 python -m pytest tests -x -q -m gpu > $out/gpu_tests.log 2>&1
rc=$?; echo "gpu tests rc=$rc" >> $out/gpu_tests.log; tail -2 $out/gpu_tests.log | cut -c1-300
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python bench.py --no-cpu-baseline > $out/bench1.json 2> $out/bench1.err; cut -c1-250 $out/bench1.json
timeout -k 10 600 python tools/bench_configs.py --only cfg4_matern52_1e6_d32 2>&1 | grep -v amdgpu.ids | cut -c1-200
