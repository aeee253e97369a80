#!/bin/bash
# session-2 check p: fuzz tools in the final state (descriptor-driven vs round-by-round; structured kernels vs the oracle; reduction kernels)
set -u
out=gpurun_out/s2p; mkdir -p $out
timeout -k 10 500 python tools/fuzz_async.py 11 120 > $out/fuzz_async.txt 2>&1; echo "rc=$?" >> $out/fuzz_async.txt; grep -v amdgpu.ids $out/fuzz_async.txt | tail -4 | cut -c1-200
timeout -k 10 500 python tools/fuzz_structured.py 7 60 > $out/fuzz_structured.txt 2>&1; echo "rc=$?" >> $out/fuzz_structured.txt; grep -v amdgpu.ids $out/fuzz_structured.txt | tail -6 | cut -c1-200
timeout -k 10 300 python tools/fuzz_reduction.py > $out/fuzz_reduction.txt 2>&1; echo "rc=$?" >> $out/fuzz_reduction.txt; grep -v amdgpu.ids $out/fuzz_reduction.txt | tail -4 | cut -c1-200
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
