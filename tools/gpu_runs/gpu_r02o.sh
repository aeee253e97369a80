#!/bin/bash
# round-2 GPU check #15: pipelined skinny GEMM -- K-split sweep vs the library, then parity + bench
set -u
out=gpurun_out/r02o; mkdir -p $out
timeout 600 python tools/bench_skinny_gemm.py > $out/skinny_q99.txt 2>&1; grep -v amdgpu.ids $out/skinny_q99.txt
timeout 600 python tools/bench_skinny_gemm.py --q 199 > $out/skinny_q199.txt 2>&1; grep -v amdgpu.ids $out/skinny_q199.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "skinny or range_finder" > $out/kernels.log 2>&1
echo "kernel tests rc=$?" >> $out/kernels.log; tail -3 $out/kernels.log
timeout 900 python bench.py --no-cpu-baseline > $out/bench1.json 2> $out/bench1.err; cut -c1-200 $out/bench1.json
