#!/bin/bash
# session-2 check y: big products with 16 rows per wave (three waves per SIMD) against 32 rows per wave
set -u
out=gpurun_out/s2y; mkdir -p $out
for v in jt2 jt1; do
  echo "== $v" >> $out/ab.txt
  BASQ_HIP_LIB=$PWD/tools/variants/$v.so timeout -k 10 200 python tools/bench_skinny_gemm.py 2>&1 | grep -v amdgpu.ids | grep "skinny_gemm" >> $out/ab.txt
done
cat $out/ab.txt
