#!/bin/bash
# round-2 GPU check #24: same-process A/Bs of engine switches in the final state
set -u
out=gpurun_out/r02x; mkdir -p $out
timeout 600 python tools/ab_engine.py BASIS_SIDE_STREAM 0 1 --reps 8 > $out/ab_side_stream.txt 2>&1; grep -v amdgpu.ids $out/ab_side_stream.txt | tail -4
timeout 600 python tools/ab_engine.py ASYNC_ROUNDS 0 1 --reps 8 > $out/ab_async_rounds.txt 2>&1; grep -v amdgpu.ids $out/ab_async_rounds.txt | tail -4
timeout 600 python tools/ab_engine.py CLASS_SUMS 0 1 --reps 6 > $out/ab_class_sums.txt 2>&1; grep -v amdgpu.ids $out/ab_class_sums.txt | tail -4
timeout 600 python tools/ab_engine.py OWN_RANGE_GEMM 0 1 --reps 8 > $out/ab_own_gemm.txt 2>&1; grep -v amdgpu.ids $out/ab_own_gemm.txt | tail -4
timeout 600 python tools/ab_engine.py LATE_CLASSES 1 2 --reps 8 > $out/ab_late_classes.txt 2>&1; grep -v amdgpu.ids $out/ab_late_classes.txt | tail -4
