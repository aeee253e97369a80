#!/bin/bash
# session-2 check g: tall-skinny GEMM with B staged through LDS -- kernel tests first, then timing
set -u
out=gpurun_out/s2g; mkdir -p $out
timeout -k 10 120 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k skinny > $out/tests_skinny.log 2>&1 || { tail -30 $out/tests_skinny.log | cut -c1-200; exit 1; }
tail -1 $out/tests_skinny.log
timeout -k 10 200 python tools/bench_skinny_gemm.py 2>&1 | grep -v amdgpu.ids | grep "skinny\|X^T" > $out/bench.txt; cat $out/bench.txt
timeout -k 10 200 python tools/bench_skinny_gemm.py --q 199 2>&1 | grep -v amdgpu.ids | grep "auto\|=  8\|= 16" > $out/bench199.txt; cat $out/bench199.txt
