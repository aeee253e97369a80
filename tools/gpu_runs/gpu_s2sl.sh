#!/bin/bash
# session-2: WSABI-M squared-covariance block sums with B^T of a 64-row group resident in LDS (7 / 8 / 4 waves per work-group) against the shipped form
set -u
out=gpurun_out/s2sl; mkdir -p $out
for v in sl7 sl8; do
  echo "== $v" >> $out/ab.txt
  BASQ_HIP_LIB=$PWD/tools/variants/$v.so timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "sq or wsabim" 2>&1 | tail -1 >> $out/ab.txt
  BASQ_HIP_LIB=$PWD/tools/variants/$v.so timeout -k 10 300 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "wsabim or noise" 2>&1 | tail -1 >> $out/ab.txt
  BASQ_HIP_LIB=$PWD/tools/variants/$v.so timeout -k 10 300 python tools/bench_configs.py --only cfg5m_wsabim_5e5 2>&1 | grep -v amdgpu.ids | cut -c1-60,150-260 >> $out/ab.txt
done
echo "== shipped form (BASQ_SQ_SLAB=0)" >> $out/ab.txt
BASQ_SQ_SLAB=0 BASQ_HIP_LIB=$PWD/tools/variants/sl7.so timeout -k 10 300 python tools/bench_configs.py --only cfg5m_wsabim_5e5 2>&1 | grep -v amdgpu.ids | cut -c1-60,150-260 >> $out/ab.txt
cat $out/ab.txt
