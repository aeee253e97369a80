#!/bin/bash
# round-2 GPU check #17: per-round read-back by event polling vs blocking stream wait (A/B), 1-rank RCCL line
set -u
out=gpurun_out/r02q; mkdir -p $out
for rep in 1 2; do
BASQ_SPIN_WAIT=0 timeout 900 python bench.py --no-cpu-baseline --steps 10 > $out/bench_block_$rep.json 2> $out/bench_block_$rep.err; echo "blocking wait: $(cut -c80-140 $out/bench_block_$rep.json)"
BASQ_SPIN_WAIT=1 timeout 900 python bench.py --no-cpu-baseline --steps 10 > $out/bench_spin_$rep.json 2> $out/bench_spin_$rep.err; echo "event polling: $(cut -c80-140 $out/bench_spin_$rep.json)"
done
BASQ_SPIN_WAIT=0 timeout 600 python tools/bench_configs.py --only cfg2_rbf_1e5 2>&1 | grep -v amdgpu.ids | cut -c1-120
BASQ_SPIN_WAIT=1 timeout 600 python tools/bench_configs.py --only cfg2_rbf_1e5 2>&1 | grep -v amdgpu.ids | cut -c1-120
BASQ_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 timeout 900 python bench.py --no-cpu-baseline > $out/bench_force_dist.json 2> $out/bench_force_dist.err; cut -c1-200 $out/bench_force_dist.json
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu > $out/parity.log 2>&1; tail -2 $out/parity.log
