#!/bin/bash
# round-2 GPU check #9: granule cluster exchange, transposed-basis projection; WSABI-M timing line
set -u
out=gpurun_out/r02i; mkdir -p $out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "cluster or nullspace or car_eliminate or project or fuzz" > $out/kernels.log 2>&1
echo "kernel tests rc=$?" >> $out/kernels.log; tail -4 $out/kernels.log
for shape in "200 400" "150 400" "256 512"; do echo "== $shape" >> $out/reduction.txt; timeout 300 python tools/bench_reduction.py $shape 2>&1 | grep -v amdgpu.ids >> $out/reduction.txt; done
cat $out/reduction.txt
timeout 2400 python -m pytest tests -x -q -m gpu > $out/gpu_tests.log 2>&1
echo "gpu tests rc=$?" >> $out/gpu_tests.log; tail -5 $out/gpu_tests.log | cut -c1-300
timeout 900 python bench.py --breakdown > $out/bench1.json 2> $out/bench1.err; cut -c1-330 $out/bench1.json
timeout 900 python bench.py > $out/bench2.json 2> $out/bench2.err; cut -c1-330 $out/bench2.json
timeout 1200 python tools/bench_configs.py > $out/configs.txt 2>&1; grep -v amdgpu.ids $out/configs.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_bench -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/prof_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_cfg4 -o cfg4 -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py --only cfg4_matern52_1e6_d32 > $GRAFT_REPO_ROOT/$out/prof_cfg4.log 2>&1
cd $GRAFT_REPO_ROOT
rm -f $out/prof_*/*trace.csv
head -12 $out/prof_bench/bench_kernel_stats.csv | cut -c1-140; head -8 $out/prof_cfg4/cfg4_kernel_stats.csv | cut -c1-140
