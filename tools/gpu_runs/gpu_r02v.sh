#!/bin/bash
# round-2 GPU check #22: how many classes of the round-1 block sums to defer behind the range finder (covers the host SVD)
set -u
out=gpurun_out/r02v; mkdir -p $out
for L in 1 2 3 1 2 3; do
BASQ_LATE_CLASSES=$L timeout 900 python bench.py --no-cpu-baseline --steps 10 > $out/bench_L${L}.json 2> $out/bench_L${L}.err; echo "late classes $L: $(cut -c80-140 $out/bench_L${L}.json)"
done
timeout 600 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | tail -2
