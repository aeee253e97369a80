#!/bin/bash
# session-2 check b: same-box A/B of block-sum variants (16x16x4 vs 4x4x4 forms; rows per wave; chain interleave)
set -u
out=gpurun_out/s2b; mkdir -p $out
for rep in 1 2; do
for v in v_old v_jt4 v_jt2 v_jt4_ilv v_jt2_ilv; do
  echo "== $v" >> $out/ab.txt
  BASQ_HIP_LIB=$PWD/tools/variants/$v.so timeout -k 10 200 python tools/bench_blocksum.py --reps 10 2>&1 | grep -v amdgpu.ids >> $out/ab.txt
done; done
cat $out/ab.txt
