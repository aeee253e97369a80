#!/bin/bash
# session-2 check m: WSABI-M squared-covariance block sums on the 4x4x4 matrix instruction: tests, then config 5 (WSABI-M) timing
set -u
out=gpurun_out/s2m; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu > $out/kernel_tests.log 2>&1 || { tail -30 $out/kernel_tests.log | cut -c1-200; exit 1; }
tail -1 $out/kernel_tests.log
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "wsabim or noise" > $out/parity_wsabim.log 2>&1 || { tail -30 $out/parity_wsabim.log | cut -c1-200; exit 1; }
tail -1 $out/parity_wsabim.log
timeout -k 10 600 python tools/bench_configs.py --only cfg5m_wsabim_5e5 > $out/cfg5m.txt 2>&1; grep -v amdgpu.ids $out/cfg5m.txt | cut -c1-250
