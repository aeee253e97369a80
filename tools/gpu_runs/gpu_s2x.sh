#!/bin/bash
# session-2 check x: threads of car_eliminate_lds_kernel at 100 x 200 (512 / 640 / 768 / 1024)
set -u
out=gpurun_out/s2x; mkdir -p $out
for v in car1024 car768 car640 car512; do
  echo "== $v" >> $out/ab.txt
  BASQ_HIP_LIB=$PWD/tools/variants/$v.so timeout -k 10 200 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "car or elimin" 2>&1 | tail -1 >> $out/ab.txt
  BASQ_HIP_LIB=$PWD/tools/variants/$v.so timeout -k 10 200 python tools/bench_reduction.py 2>&1 | grep "car_eliminate" >> $out/ab.txt
done
cat $out/ab.txt
