#!/bin/bash
# session-2 check v: threads of the panel Cholesky at q = 99 (128 / 256 / 512 / 1024)
set -u
out=gpurun_out/s2v; mkdir -p $out
for v in ch1024 ch512 ch256 ch128; do
  echo "== $v" >> $out/ab.txt
  BASQ_HIP_LIB=$PWD/tools/variants/$v.so timeout -k 10 200 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "chol" 2>&1 | tail -1 >> $out/ab.txt
  BASQ_HIP_LIB=$PWD/tools/variants/$v.so timeout -k 10 200 python tools/bench_reduction.py 2>&1 | grep "chol_factor\|trsm" >> $out/ab.txt
done
cat $out/ab.txt
