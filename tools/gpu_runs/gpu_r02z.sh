#!/bin/bash
# round-2 GPU check #26: fuzzers through the production path (descriptor-driven rounds)
set -u
out=gpurun_out/r02z; mkdir -p $out
timeout 1500 python tools/fuzz_async.py 1 60 > $out/fuzz_async_seed1.txt 2>&1; grep -v amdgpu.ids $out/fuzz_async_seed1.txt | tail -5
timeout 1500 python tools/fuzz_async.py 2 60 > $out/fuzz_async_seed2.txt 2>&1; grep -v amdgpu.ids $out/fuzz_async_seed2.txt | tail -5
timeout 1500 python tools/fuzz_structured.py 11 120 > $out/fuzz_structured_seed11.txt 2>&1; grep -v amdgpu.ids $out/fuzz_structured_seed11.txt | tail -6
timeout 900 python tools/fuzz_reduction.py --cases 200 --seed 5 > $out/fuzz_reduction_seed5.txt 2>&1; grep -v amdgpu.ids $out/fuzz_reduction_seed5.txt | tail -4
