#!/bin/bash
# session-2 check c: tall-skinny GEMM on v_mfma_f64_4x4x4_4b, same-box A/B against the 16x16x4 kernel
set -u
out=gpurun_out/s2c; mkdir -p $out; rm -f $out/ab.txt
for v in sk_old sk_dpp_jt2; do
  echo "== $v" >> $out/ab.txt
  BASQ_HIP_LIB=$PWD/tools/variants/$v.so timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k skinny 2>&1 | tail -1 >> $out/ab.txt
  BASQ_HIP_LIB=$PWD/tools/variants/$v.so timeout -k 10 200 python tools/bench_skinny_gemm.py 2>&1 | grep -v amdgpu.ids | grep "skinny\|X^T" | grep "auto\|= 16\|=  3\|=  8\|= 62\|= 26\|= 64" >> $out/ab.txt
  BASQ_HIP_LIB=$PWD/tools/variants/$v.so timeout -k 10 200 python tools/bench_skinny_gemm.py --q 199 2>&1 | grep -v amdgpu.ids | grep "auto\|=  8\|= 16" >> $out/ab.txt
done
cat $out/ab.txt
