#!/bin/bash
# round-2 GPU check #13: in-kernel phase stamps of the one-CU reductions (tools/ns_prof, -DBASQ_NS_PROF build)
set -u
out=gpurun_out/r02m; mkdir -p $out
timeout 300 ./tools/ns_prof 100 200 > $out/ns_prof_100x200.txt 2>&1; cat $out/ns_prof_100x200.txt | grep -v amdgpu.ids | cut -c1-150
timeout 900 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err; cut -c1-200 $out/bench.json
