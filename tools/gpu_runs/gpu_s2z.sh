#!/bin/bash
# session-2 final check: full GPU suite, bench x2 (first with the CPU baseline), configs, rocprof stats of the bench / config 4 / config 5 WSABI-M,
# one-rank RCCL line, idle gaps
set -u
out=gpurun_out/s2z; mkdir -p $out
timeout -k 10 1500 python -m pytest tests -x -q -m gpu > $out/gpu_tests.log 2>&1
rc=$?; echo "gpu tests rc=$rc" >> $out/gpu_tests.log; tail -3 $out/gpu_tests.log | cut -c1-300
[ $rc -eq 0 ] || exit 1
timeout -k 10 1500 python bench.py --breakdown > $out/bench1.json 2> $out/bench1.err; cut -c1-330 $out/bench1.json
timeout -k 10 900 python bench.py --no-cpu-baseline > $out/bench2.json 2> $out/bench2.err; cut -c1-200 $out/bench2.json
timeout -k 10 900 python tools/bench_configs.py > $out/configs.txt 2>&1; grep -v amdgpu.ids $out/configs.txt | cut -c1-330
BASQ_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 timeout -k 10 900 python bench.py --no-cpu-baseline > $out/bench_force_dist.json 2> $out/bench_force_dist.err; cut -c1-200 $out/bench_force_dist.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_bench -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/prof_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_cfg4 -o cfg4 -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py --only cfg4_matern52_1e6_d32 > $GRAFT_REPO_ROOT/$out/prof_cfg4.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_gaps.py $(ls $out/prof_bench/*kernel_trace.csv | head -1) > $out/trace_gaps_bench.txt 2>&1; head -6 $out/trace_gaps_bench.txt | cut -c1-200
rm -f $out/prof_*/*trace.csv
head -14 $out/prof_bench/bench_kernel_stats.csv | cut -c1-150
