#!/bin/bash
# session-2 check r: committed state after the apply-kernel change: full GPU suite + bench
set -u
out=gpurun_out/s2r; mkdir -p $out
timeout -k 10 1500 python -m pytest tests -x -q -m gpu > $out/gpu_tests.log 2>&1
rc=$?; echo "gpu tests rc=$rc" >> $out/gpu_tests.log; tail -3 $out/gpu_tests.log | cut -c1-300
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python bench.py --no-cpu-baseline > $out/bench1.json 2> $out/bench1.err; cut -c1-330 $out/bench1.json
timeout -k 10 600 python bench.py --no-cpu-baseline > $out/bench2.json 2> $out/bench2.err; cut -c1-200 $out/bench2.json
