#!/bin/bash
# session-2 check f: full GPU suite, headline bench and the other configurations with the 4x4x4 tall-skinny GEMM in the library
set -u
out=gpurun_out/s2f; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k skinny > $out/tests_skinny.log 2>&1 || { tail -20 $out/tests_skinny.log | cut -c1-200; exit 1; }
tail -1 $out/tests_skinny.log
timeout -k 10 1500 python -m pytest tests -x -q -m gpu > $out/tests.log 2>&1
rc=$?; echo "tests rc=$rc" >> $out/tests.log; tail -3 $out/tests.log | cut -c1-300
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python bench.py --no-cpu-baseline --breakdown > $out/bench1.json 2> $out/bench1.err; cut -c1-330 $out/bench1.json
timeout -k 10 600 python tools/bench_configs.py > $out/configs.txt 2>&1; grep -v amdgpu.ids $out/configs.txt | cut -c1-250
