#!/bin/bash
# round-2 GPU check #8: XCD-aware block-sum map, 7-tile projection GEMM, range finder on a side stream (A/B), PMC traffic
set -u
out=gpurun_out/r02h; mkdir -p $out
timeout 2400 python -m pytest tests -x -q -m gpu > $out/gpu_tests.log 2>&1
echo "gpu tests rc=$?" >> $out/gpu_tests.log; tail -5 $out/gpu_tests.log | cut -c1-300
timeout 300 python tools/bench_blocksum.py > $out/blocksum.txt 2>&1; grep -v amdgpu.ids $out/blocksum.txt
timeout 600 python tools/ab_engine.py BASIS_SIDE_STREAM 0 1 --reps 6 --visits 3 > $out/ab_side_stream.txt 2>&1; grep -v amdgpu.ids $out/ab_side_stream.txt
timeout 900 python bench.py --breakdown > $out/bench1.json 2> $out/bench1.err; cut -c1-330 $out/bench1.json
timeout 900 python tools/bench_configs.py > $out/configs.txt 2>&1; grep -v amdgpu.ids $out/configs.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_bench -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/prof_bench.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$out/pmc_fetch -o fetch -- python3 $GRAFT_REPO_ROOT/tools/bench_blocksum.py --reps 2 > $GRAFT_REPO_ROOT/$out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$out/pmc_write -o write -- python3 $GRAFT_REPO_ROOT/tools/bench_blocksum.py --reps 2 > $GRAFT_REPO_ROOT/$out/pmc_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/$out/pmc_pipe -o pipe -- python3 $GRAFT_REPO_ROOT/tools/bench_blocksum.py --reps 2 > $GRAFT_REPO_ROOT/$out/pmc_pipe.log 2>&1
cd $GRAFT_REPO_ROOT
rm -f $out/prof_*/*trace.csv
head -12 $out/prof_bench/bench_kernel_stats.csv | cut -c1-140
for d in pmc_fetch pmc_write pmc_pipe; do f=$(find $out/$d -name "*counter_collection.csv" | head -1); echo $f; grep blocksum $f | tail -4 | cut -c60-260; done
