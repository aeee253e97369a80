#!/bin/bash
# round-2 GPU check #18: descriptor-driven rounds (no host wait per round): kernel parity, goldens, bench, gaps
set -u
out=gpurun_out/r02r; mkdir -p $out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "descriptor or blocksum" > $out/kernels.log 2>&1
echo "kernel tests rc=$?" >> $out/kernels.log; tail -15 $out/kernels.log | cut -c1-200
timeout 1500 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "descriptor" > $out/parity_async.log 2>&1
echo "async parity rc=$?" >> $out/parity_async.log; tail -15 $out/parity_async.log | cut -c1-200
timeout 900 python bench.py --no-cpu-baseline --steps 10 > $out/bench1.json 2> $out/bench1.err; cut -c1-330 $out/bench1.json; tail -3 $out/bench1.err | cut -c1-300
timeout 900 python bench.py --no-cpu-baseline --steps 10 > $out/bench2.json 2> $out/bench2.err; cut -c1-200 $out/bench2.json
timeout 1200 python tools/bench_configs.py > $out/configs.txt 2>&1; grep -v amdgpu.ids $out/configs.txt | cut -c1-130
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_bench -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/prof_bench.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_gaps.py $(ls $out/prof_bench/*kernel_trace.csv | head -1) > $out/trace_gaps_bench.txt 2>&1; head -8 $out/trace_gaps_bench.txt | cut -c1-200
rm -f $out/prof_*/*trace.csv
timeout 2400 python -m pytest tests -x -q -m gpu > $out/gpu_tests.log 2>&1
echo "gpu tests rc=$?" >> $out/gpu_tests.log; tail -5 $out/gpu_tests.log | cut -c1-300
