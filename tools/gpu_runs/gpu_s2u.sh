#!/bin/bash
# session-2 check u: triangular solve with the factor in LDS, finalize sums with loads in flight: full suite, bench, range-finder launch sequence
set -u
out=gpurun_out/s2u; mkdir -p $out
timeout -k 10 1500 python -m pytest tests -x -q -m gpu > $out/gpu_tests.log 2>&1
rc=$?; echo "gpu tests rc=$rc" >> $out/gpu_tests.log; tail -2 $out/gpu_tests.log | cut -c1-300
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python bench.py --no-cpu-baseline > $out/bench1.json 2> $out/bench1.err; cut -c1-250 $out/bench1.json
timeout -k 10 600 python bench.py --no-cpu-baseline > $out/bench2.json 2> $out/bench2.err; cut -c1-250 $out/bench2.json
bash tools/gpu_runs/gpu_s2k.sh > /dev/null 2>&1; grep "trsm\|chol\|finalize" gpurun_out/s2k/seq.txt | head -5
