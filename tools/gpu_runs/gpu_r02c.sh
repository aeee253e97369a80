#!/bin/bash
# round-2 GPU check #3: cluster-kernel phase profile, kernel parity tests, block-sum form A/B
set -u
out=gpurun_out/r02c; mkdir -p $out
./tools/ns_prof2 100 200 > $out/ns_prof_100x200.txt 2>&1
./tools/ns_prof2 200 400 > $out/ns_prof_200x400.txt 2>&1
head -40 $out/ns_prof_100x200.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "nullspace or car_eliminate or cluster or fuzz or blocksum or matvec" > $out/kernels.log 2>&1
echo "kernel tests rc=$?" >> $out/kernels.log
tail -5 $out/kernels.log
timeout 300 python tools/bench_blocksum.py > $out/blocksum_default.txt 2>&1; cat $out/blocksum_default.txt | grep -v amdgpu.ids
for cfg in "3 1 8" "2 2 8" "2 1 16" "2 1 4"; do
  set -- $cfg
  python - > $out/build_$1_$2_$3.log 2>&1 <<PY
from basq_amd import _build
_build.build(force=True, verbose=False, defines={"BASQ_LDS_WAVES": $1, "BASQ_LDS_GROUP": $2, "BASQ_LDS_ST": $3}, out="/tmp/libbasq_v.so")
PY
  echo "== waves=$1 group=$2 ST=$3" >> $out/blocksum_variants.txt
  BASQ_HIP_LIB=/tmp/libbasq_v.so timeout 300 python tools/bench_blocksum.py 2>&1 | grep -v amdgpu.ids >> $out/blocksum_variants.txt
done
cat $out/blocksum_variants.txt
timeout 300 python tools/bench_blocksum.py --R 250000 --m 10000 >> $out/blocksum_default.txt 2>&1
timeout 300 python tools/bench_blocksum.py --R 1000000 --m 10000 --d 32 --n 200 --family matern52 >> $out/blocksum_default.txt 2>&1
tail -8 $out/blocksum_default.txt | grep -v amdgpu.ids
