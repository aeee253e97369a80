#!/bin/bash
# round-2 GPU check #7: class MESSAGES across rounds; PMC traffic of the round-1 block-sum launch
set -u
out=gpurun_out/r02g; mkdir -p $out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "regroup or project or blocksum" > $out/kernels.log 2>&1
echo "kernel tests rc=$?" >> $out/kernels.log; tail -4 $out/kernels.log
timeout 2400 python -m pytest tests -x -q -m gpu > $out/gpu_tests.log 2>&1
echo "gpu tests rc=$?" >> $out/gpu_tests.log; tail -6 $out/gpu_tests.log | cut -c1-300
timeout 900 python bench.py --breakdown > $out/bench1.json 2> $out/bench1.err; tail -3 $out/bench1.err | cut -c1-700; cut -c1-330 $out/bench1.json
timeout 900 python tools/bench_configs.py > $out/configs.txt 2>&1; grep -v amdgpu.ids $out/configs.txt
timeout 300 python tools/bench_blocksum.py > $out/blocksum.txt 2>&1; grep -v amdgpu.ids $out/blocksum.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_bench -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/prof_bench.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$out/pmc_fetch -o fetch -- python3 $GRAFT_REPO_ROOT/tools/bench_blocksum.py --reps 2 > $GRAFT_REPO_ROOT/$out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$out/pmc_write -o write -- python3 $GRAFT_REPO_ROOT/tools/bench_blocksum.py --reps 2 > $GRAFT_REPO_ROOT/$out/pmc_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/$out/pmc_pipe -o pipe -- python3 $GRAFT_REPO_ROOT/tools/bench_blocksum.py --reps 2 > $GRAFT_REPO_ROOT/$out/pmc_pipe.log 2>&1
cd $GRAFT_REPO_ROOT
rm -f $out/prof_*/*trace.csv
head -14 $out/prof_bench/bench_kernel_stats.csv | cut -c1-140
for d in pmc_fetch pmc_write pmc_pipe; do f=$(find $out/$d -name "*counter_collection.csv" | head -1); echo $f; grep blocksum $f | head -6 | cut -c1-260; done
