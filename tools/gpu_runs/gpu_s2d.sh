#!/bin/bash
# session-2 check d: the tall-skinny GEMM's loads without its products (probe build), against the full kernel
set -u
out=gpurun_out/s2d; mkdir -p $out; rm -f $out/ab.txt
for v in sk_probe sk_probe_noA sk_probe_noB; do
  for m in 10000; do
  echo "== $v m=$m" >> $out/ab.txt
  BASQ_HIP_LIB=$PWD/tools/variants/$v.so timeout -k 10 200 python tools/bench_skinny_gemm.py --m $m 2>&1 | grep -v amdgpu.ids | grep "skinny_gemm\|GFLOP" | grep "GFLOP\|auto\|= 16\|=  8\|=  3 " >> $out/ab.txt
  done
done
cat $out/ab.txt
