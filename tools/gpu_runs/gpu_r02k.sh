#!/bin/bash
# round-2 GPU check #11: fused WSABI-M (basq_blocksum_sq_f64) parity + timing (JT 4 vs 2), host profile, GPU idle gaps
set -u
out=gpurun_out/r02k; mkdir -p $out
timeout 900 python -m pytest tests -x -q -m gpu -k "blocksum_sq or wsabim or wsabi" > $out/wsabi_tests.log 2>&1
echo "wsabi tests rc=$?" >> $out/wsabi_tests.log; tail -4 $out/wsabi_tests.log | cut -c1-300
timeout 600 python tools/bench_configs.py --only cfg5m_wsabim_5e5 > $out/cfg5m_jt4.txt 2>&1; grep -v amdgpu.ids $out/cfg5m_jt4.txt | cut -c1-330
BASQ_SQ_JT=2 timeout 600 python tools/bench_configs.py --only cfg5m_wsabim_5e5 > $out/cfg5m_jt2.txt 2>&1; grep -v amdgpu.ids $out/cfg5m_jt2.txt | cut -c1-330
timeout 600 python tools/host_profile.py > $out/host_profile.txt 2>&1; grep -v amdgpu.ids $out/host_profile.txt | head -45 | cut -c1-200
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_bench -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/prof_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_cfg5m -o cfg5m -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py --only cfg5m_wsabim_5e5 > $GRAFT_REPO_ROOT/$out/prof_cfg5m.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_gaps.py $(ls $out/prof_bench/*kernel_trace.csv | head -1) > $out/trace_gaps_bench.txt 2>&1; cat $out/trace_gaps_bench.txt | cut -c1-200
rm -f $out/prof_*/*trace.csv
head -8 $out/prof_cfg5m/cfg5m_kernel_stats.csv | cut -c1-150
timeout 2400 python -m pytest tests -x -q -m gpu > $out/gpu_tests.log 2>&1
echo "gpu tests rc=$?" >> $out/gpu_tests.log; tail -5 $out/gpu_tests.log | cut -c1-300
