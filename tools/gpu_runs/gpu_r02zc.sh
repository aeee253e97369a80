#!/bin/bash
# round-2 GPU check #30: bench.py after the roofline-JSON additions (reference pair count), smoke
set -u
out=gpurun_out/r02zc; mkdir -p $out
timeout 1500 python bench.py > $out/bench1.json 2> $out/bench1.err; tail -1 $out/bench1.json | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['roofline']['reference_pairs_per_batch'], d['roofline']['pairs_per_batch'], d['roofline']['whole_batch_TFLOPs_by_reference_count'], d['cpu_baseline']['value'])"
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -1
