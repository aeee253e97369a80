#!/bin/bash
# One parameterised runner for everything this repo hands to `gpurun` (replaces the per-call scripts of rounds 1-2, whose
# history is in git: tools/gpu_runs/ up to commit 98fa8bf).
#
#   gpurun --timeout 1200 -- 'bash tools/gpu_jobs.sh <tag> <job> [<job> ...]'
#
# Every job writes under gpurun_out/<tag>/ and prints a short tail; a job that fails or times out stops the call (no GPU
# step is started after a killed one).  What was copied from there into profiles/ is listed in profiles/README.md.
set -u
tag=$1; shift
out=gpurun_out/$tag; mkdir -p "$out"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp

note() { echo "[$(date +%H:%M:%S)] $*" | tee -a "$out/jobs.log"; }

job_tests()       { timeout -k 10 1100 python -m pytest tests -x -q -m gpu > "$out/gpu_tests.log" 2>&1; rc=$?; tail -4 "$out/gpu_tests.log" | cut -c1-300; return $rc; }
job_tests_fast()  { timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "not cfg3 and not cfg4 and not cfg5 and not full_size" > "$out/gpu_tests_fast.log" 2>&1; rc=$?; tail -4 "$out/gpu_tests_fast.log" | cut -c1-300; return $rc; }
job_smoke()       { timeout -k 10 300 python __graft_entry__.py smoke > "$out/smoke.log" 2>&1; rc=$?; tail -2 "$out/smoke.log"; return $rc; }
job_bench()       { timeout -k 10 900 python bench.py --breakdown > "$out/bench.json" 2> "$out/bench.err"; rc=$?; cut -c1-400 "$out/bench.json"; return $rc; }
job_bench20()     { timeout -k 10 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > "$out/bench20.json" 2> "$out/bench20.err"; rc=$?; cut -c1-400 "$out/bench20.json"; return $rc; }
job_bench_nocpu() { timeout -k 10 600 python bench.py --no-cpu-baseline > "$out/bench_nocpu.json" 2> "$out/bench_nocpu.err"; rc=$?; cut -c1-400 "$out/bench_nocpu.json"; return $rc; }
job_bench_dist1() { BASQ_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 timeout -k 10 600 python bench.py --no-cpu-baseline > "$out/bench_force_dist.json" 2> "$out/bench_force_dist.err"; rc=$?; cut -c1-300 "$out/bench_force_dist.json"; return $rc; }
job_configs()     { timeout -k 10 900 python tools/bench_configs.py > "$out/configs.txt" 2>&1; rc=$?; grep -v amdgpu.ids "$out/configs.txt" | cut -c1-330; return $rc; }
job_concurrent()  { timeout -k 10 600 python tools/exp_concurrent.py --threads 2 --launches > "$out/concurrent.txt" 2>&1; rc=$?; grep -v amdgpu.ids "$out/concurrent.txt" | cut -c1-200; return $rc; }
job_concurrent3() { timeout -k 10 600 python tools/exp_concurrent.py --threads 3 > "$out/concurrent3.txt" 2>&1; rc=$?; grep -v amdgpu.ids "$out/concurrent3.txt" | cut -c1-200; return $rc; }
job_many()        { timeout -k 10 600 python tools/bench_many.py > "$out/many.txt" 2>&1; rc=$?; grep -v amdgpu.ids "$out/many.txt" | cut -c1-250; return $rc; }
job_many_small()  { ( timeout -k 10 300 python tools/bench_many.py --case rbf_2e4_defaults --batches 48 --inflight 2,4,6 2>&1 | grep -v amdgpu.ids; timeout -k 10 300 python tools/bench_many.py --case cfg2_rbf_1e5 --batches 32 --inflight 2,4 2>&1 | grep -v amdgpu.ids ) > "$out/many_small.txt" 2>&1; rc=$?; cut -c1-250 "$out/many_small.txt"; return $rc; }
job_many_cfg4()   { timeout -k 10 600 python tools/bench_many.py --case cfg4_matern52_1e6_d32 --batches 8 --inflight 2,3 > "$out/many_cfg4.txt" 2>&1; rc=$?; grep -v amdgpu.ids "$out/many_cfg4.txt" | cut -c1-250; return $rc; }
job_many_cfg5m()  { timeout -k 10 600 python tools/bench_many.py --case cfg5m_wsabim_5e5 --batches 8 --inflight 2 > "$out/many_cfg5m.txt" 2>&1; rc=$?; grep -v amdgpu.ids "$out/many_cfg5m.txt" | cut -c1-250; return $rc; }
# A/B of a tuning-variant build of the library (BASQ_HIP_LIB): block-sum micro-benchmark A B A B, then the golden parity tests on B
job_ab_lib()      { B=${AB_LIB:-basq_amd/csrc/libbasq_hip_jt2.so}; for i in 1 2 3; do for lib in basq_amd/csrc/libbasq_hip.so $B; do echo "== $lib" >> "$out/ab_lib.txt"; BASQ_HIP_LIB=$PWD/$lib timeout -k 10 120 python tools/bench_blocksum.py --reps 6 2>&1 | grep -v amdgpu.ids >> "$out/ab_lib.txt" || return 1; BASQ_HIP_LIB=$PWD/$lib timeout -k 10 120 python tools/bench_blocksum.py --family matern52 --d 32 --n 200 --reps 3 2>&1 | grep -v amdgpu.ids >> "$out/ab_lib.txt" || return 1; done; done; cat "$out/ab_lib.txt"; BASQ_HIP_LIB=$PWD/$B timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "golden or fuzz or moment" > "$out/ab_lib_tests.log" 2>&1; rc=$?; tail -3 "$out/ab_lib_tests.log"; return $rc; }
job_fuzz()        { ( timeout -k 10 500 python tools/fuzz_async.py 5 80 2>&1 | grep -v amdgpu.ids | tail -4; timeout -k 10 500 python tools/fuzz_structured.py 11 120 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" | tail -8 ) > "$out/fuzz.txt" 2>&1; cat "$out/fuzz.txt"; }
job_ab_lib_bench() { B=${AB_LIB:-basq_amd/csrc/libbasq_hip_jt2.so}; for i in 1 2; do for lib in basq_amd/csrc/libbasq_hip.so $B; do echo "== $lib" >> "$out/ab_lib_bench.txt"; BASQ_HIP_LIB=$PWD/$lib timeout -k 10 200 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-concurrent 2>/dev/null | cut -c1-260 >> "$out/ab_lib_bench.txt" || return 1; done; done; cat "$out/ab_lib_bench.txt"; }
job_ab_sqpf()     { for i in 1 2; do for lib in libbasq_hip.so libbasq_hip_pf2.so libbasq_hip_pf3.so; do echo "== $lib" >> "$out/ab_sqpf.txt"; BASQ_HIP_LIB=$PWD/basq_amd/csrc/$lib timeout -k 10 200 python tools/bench_configs.py --only cfg5m_wsabim_5e5 --reps 4 2>&1 | grep -v amdgpu.ids | cut -c1-100 >> "$out/ab_sqpf.txt" || return 1; done; done; cat "$out/ab_sqpf.txt"; BASQ_HIP_LIB=$PWD/basq_amd/csrc/libbasq_hip_pf2.so timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "wsabim or blocksum_sq" 2>&1 | tail -2; }
job_ab_idle()     { B=${AB_LIB:-basq_amd/csrc/libbasq_hip_bpf2.so}; for i in 1 2; do for lib in basq_amd/csrc/libbasq_hip.so $B; do echo "== $lib" >> "$out/ab_idle.txt"; BASQ_HIP_LIB=$PWD/$lib timeout -k 10 200 python tools/idle_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-200 >> "$out/ab_idle.txt" || return 1; done; done; cat "$out/ab_idle.txt"; }
job_fuzz_more()   { timeout -k 10 1100 python tools/fuzz_more.py --seeds ${FUZZ_SEEDS:-1 2 3} --count ${FUZZ_COUNT:-100} > "$out/fuzz_more.txt" 2>&1; rc=$?; grep -v amdgpu.ids "$out/fuzz_more.txt" | cut -c1-400 | tail -40; return $rc; }
job_ab_spin()     { timeout -k 10 400 python tools/ab_engine.py SPIN_WAIT 0 1 --reps 10 --visits 3 2>&1 | grep -v amdgpu.ids > "$out/ab_spin_wait.txt"; rc=$?; cat "$out/ab_spin_wait.txt"; return $rc; }
job_ab_car()      { ( for r in 1 0 1 0; do echo "== BASQ_CAR_RING=$r"; BASQ_CAR_RING=$r timeout -k 10 120 python tools/bench_reduction.py 100 200 --reps 200 2>&1 | grep -E "car_eliminate|nullspace "; done; for sh in "50 100" "31 62" "100 150"; do for r in 1 0; do echo "== $sh BASQ_CAR_RING=$r"; BASQ_CAR_RING=$r timeout -k 10 120 python tools/bench_reduction.py $sh --reps 100 2>&1 | grep -E "car_eliminate"; done; done ) > "$out/ab_car_ring.txt" 2>&1; rc=$?; cat "$out/ab_car_ring.txt"; return $rc; }
job_car_prof()    { /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -Wno-unused-function -DBASQ_NS_PROF tools/car_prof.hip -o /tmp/car_prof > "$out/car_prof_build.log" 2>&1 && timeout -k 10 120 /tmp/car_prof 100 200 > "$out/car_prof.txt" 2>&1; rc=$?; cat "$out/car_prof.txt" | cut -c1-200; return $rc; }
job_ns_prof()     { /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -Wno-unused-function -DBASQ_NS_PROF tools/ns_prof.hip -o /tmp/ns_prof > "$out/ns_prof_build.log" 2>&1 && timeout -k 10 120 /tmp/ns_prof 100 200 > "$out/ns_prof.txt" 2>&1; rc=$?; head -40 "$out/ns_prof.txt" | cut -c1-200; return $rc; }
job_ab_copy_small() { ( for r in 1 2; do for v in 0 1; do for c in cfg2_rbf_1e5 rbf_2e4_defaults; do timeout -k 10 200 python tools/bench_many.py --case $c --batches 24 --inflight 4 --set RAND_COPY_STREAM=$v 2>&1 | grep -v amdgpu.ids; done; done; done ) > "$out/ab_copy_small.txt" 2>&1; rc=$?; cat "$out/ab_copy_small.txt" | cut -c1-150; return $rc; }
job_tests_ns()    { timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "nullspace or car_eliminate" > "$out/gpu_tests_ns.log" 2>&1; rc=$?; tail -6 "$out/gpu_tests_ns.log" | cut -c1-300; return $rc; }
job_bench_red()   { ( for sh in "100 200" "50 100" "31 62"; do timeout -k 10 120 python tools/bench_reduction.py $sh --reps 200 2>&1 | grep -E "nullspace|car_eliminate"; done ) > "$out/bench_reduction.txt" 2>&1; rc=$?; cat "$out/bench_reduction.txt"; return $rc; }
job_ab_gring()    { ( for sh in "200 400" "150 300"; do for r in 1 0 1 0; do echo "== $sh BASQ_CAR_GRING=$r"; BASQ_CAR_GRING=$r timeout -k 10 120 python tools/bench_reduction.py $sh --reps 100 2>&1 | grep -E "car_eliminate"; done; done ) > "$out/ab_car_gring.txt" 2>&1; rc=$?; cat "$out/ab_car_gring.txt"; return $rc; }
job_cfg2_reps()   { timeout -k 10 300 python tools/bench_configs.py --only cfg2_rbf_1e5 --reps 30 2>&1 | grep -v amdgpu.ids > "$out/cfg2_reps.txt"; rc=$?; cut -c1-200 "$out/cfg2_reps.txt"; return $rc; }
job_gring_variants() { ( for e in "" _gr8x8 _gr4x16 _gr3x8 _gr2x8; do echo "== lib$e"; for sh in "200 400"; do BASQ_HIP_LIB=$PWD/basq_amd/csrc/libbasq_hip$e.so timeout -k 10 120 python tools/bench_reduction.py $sh --reps 100 2>&1 | grep -E "car_eliminate"; done; done ) > "$out/gring_variants.txt" 2>&1; rc=$?; cat "$out/gring_variants.txt"; return $rc; }
job_ring_variants() { ( for e in "" _rg9x12 _rg10x10 _rg13x8 _rg8x13; do echo "== lib$e"; BASQ_HIP_LIB=$PWD/basq_amd/csrc/libbasq_hip$e.so timeout -k 10 120 python tools/bench_reduction.py 100 200 --reps 200 2>&1 | grep -E "car_eliminate"; done ) > "$out/ring_variants.txt" 2>&1; rc=$?; cat "$out/ring_variants.txt"; return $rc; }
job_tests_car()   { timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "car_eliminate or nullspace_then" > "$out/gpu_tests_car.log" 2>&1; rc=$?; tail -15 "$out/gpu_tests_car.log" | cut -c1-300; return $rc; }
job_ab_copy()     { timeout -k 10 400 python tools/ab_engine.py RAND_COPY_STREAM 0 1 --reps 10 --visits 3 2>&1 | grep -v amdgpu.ids > "$out/ab_rand_copy_stream.txt"; rc=$?; cat "$out/ab_rand_copy_stream.txt"; return $rc; }
job_ab_late()     { timeout -k 10 400 python tools/ab_engine.py LATE_CLASSES 2 3 0 4 5 --reps 8 --visits 3 2>&1 | grep -v amdgpu.ids > "$out/ab_late_classes.txt"; rc=$?; cat "$out/ab_late_classes.txt"; return $rc; }
job_stall()       { for g in default collect freeze; do timeout -k 10 300 python tools/stall_probe.py --steps 60 --gc $g > "$out/stall_$g.txt" 2>&1 || return 1; grep -v amdgpu.ids "$out/stall_$g.txt" | cut -c1-300 | head -3; done; }
job_trace_small() { prof small tools/bench_many.py --case rbf_2e4_defaults --batches 12 --inflight "" && python tools/trace_batch.py "$(ls $out/prof_small/*kernel_trace.csv | head -1)" > "$out/trace_batch_2e4.txt" 2>&1; rc=$?; tail -34 "$out/trace_batch_2e4.txt" | cut -c1-160; return $rc; }
job_sq()          { timeout -k 10 200 python tools/bench_blocksum_sq.py > "$out/blocksum_sq.txt" 2>&1; rc=$?; grep -v amdgpu.ids "$out/blocksum_sq.txt"; return $rc; }
job_sober_tut()   { timeout -k 10 600 python tools/bench_sober_tutorial.py > "$out/sober_tutorial.txt" 2>&1; rc=$?; grep -v amdgpu.ids "$out/sober_tutorial.txt" | cut -c1-330; return $rc; }
job_dense_sweep() { for S in 400 200; do for ns in 0 1 2 3 4 5; do BASQ_DBS_NS=$ns timeout -k 10 120 python tools/bench_dense_blocksum.py --S $S 2>&1 | grep -v amdgpu.ids >> "$out/dense_sweep.txt" || return 1; done; done; cat "$out/dense_sweep.txt"; }
job_dense_fresh() { for nt in 0 1; do for fr in "" "--fresh"; do BASQ_DBS_NT=$nt timeout -k 10 120 python tools/bench_dense_blocksum.py --S 400 $fr 2>&1 | grep -v amdgpu.ids >> "$out/dense_fresh.txt" || return 1; done; done; BASQ_DBS_NT=1 timeout -k 10 200 python -m pytest tests/test_kernels_gpu.py -q -k dense_blocksum 2>&1 | tail -1 >> "$out/dense_fresh.txt"; cat "$out/dense_fresh.txt"; }
job_sober_phases() { timeout -k 10 300 python tools/sober_phases.py > "$out/sober_phases.txt" 2>&1; rc=$?; grep -v amdgpu.ids "$out/sober_phases.txt" | cut -c1-400; return $rc; }
job_opaque_cfg4() { timeout -k 10 1100 python tools/bench_opaque_cfg4.py > "$out/opaque_cfg4.txt" 2>&1; rc=$?; grep -v amdgpu.ids "$out/opaque_cfg4.txt" | cut -c1-250; return $rc; }

# rocprofv3 kernel statistics of a python command: prof <name> <script> [args...]  (program itself after `--`, run from /tmp)
prof() {
    local name=$1; shift
    ( cd /tmp && timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$out/prof_$name" -o "$name" -- python3 "$ROOT/$1" "${@:2}" > "$ROOT/$out/prof_$name.log" 2>&1 )
    local rc=$?
    head -16 "$out/prof_$name/${name}_kernel_stats.csv" 2>/dev/null | cut -c1-160
    return $rc
}
job_prof_bench()  { prof bench bench.py --steps 10 --warmup 2 --no-cpu-baseline --plain && python tools/trace_gaps.py "$(ls $out/prof_bench/*kernel_trace.csv | head -1)" > "$out/trace_gaps_bench.txt" 2>&1; rc=$?; head -6 "$out/trace_gaps_bench.txt" | cut -c1-200; return $rc; }
job_trace_round() { prof round bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-concurrent --no-roofline-batch && python tools/trace_round.py "$(ls $out/prof_round/*kernel_trace.csv | head -1)" --first 2 --count 2 > "$out/trace_round.txt" 2>&1; rc=$?; cat "$out/trace_round.txt" | cut -c1-140; return $rc; }
job_prof_many()   { prof many tools/bench_many.py --batches 12 --inflight 2 --pipelined-only && python tools/trace_overlap.py --tail-fraction 0.7 "$(ls $out/prof_many/*kernel_trace.csv | head -1)" > "$out/trace_overlap_many.txt" 2>&1; rc=$?; head -30 "$out/trace_overlap_many.txt" | cut -c1-200; return $rc; }
job_prof_concurrent() { prof conc tools/exp_concurrent.py --threads 2 --batches 6 && python tools/trace_overlap.py "$(ls $out/prof_conc/*kernel_trace.csv | head -1)" > "$out/trace_overlap_conc.txt" 2>&1; rc=$?; head -30 "$out/trace_overlap_conc.txt" | cut -c1-200; return $rc; }
job_prof_opaque() { prof opaque tools/bench_opaque_cfg4.py --reps 1; }
job_prof_cfg4()   { prof cfg4 tools/bench_configs.py --only cfg4_matern52_1e6_d32; }
job_prof_cfg5m()  { prof cfg5m tools/bench_configs.py --only cfg5m_wsabim_5e5; }
job_prof_cfg2()   { prof cfg2 tools/bench_configs.py --only cfg2_rbf_1e5 --reps 6 && python tools/trace_batch.py "$(ls $out/prof_cfg2/*kernel_trace.csv | head -1)" > "$out/trace_batch_cfg2.txt" 2>&1; tail -40 "$out/trace_batch_cfg2.txt" | cut -c1-150; }
job_drop_traces() { rm -f $out/prof_*/*trace.csv; }

# hardware counters of the dominant kernel, one pass per counter group (never combined with a trace domain)
pmc() {
    local name=$1 ctrs=$2
    ( cd /tmp && timeout -k 10 600 rocprofv3 --pmc $ctrs --output-format csv -d "$ROOT/$out/pmc_$name" -o "$name" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-roofline-batch > "$ROOT/$out/pmc_$name.log" 2>&1 )
}
# the dense block sums under the FETCH_SIZE / WRITE_SIZE counters (their roof is bytes: 8 B per pair, read once)
pmc_dense() {
    local name=$1 ctrs=$2
    ( cd /tmp && timeout -k 10 300 rocprofv3 --pmc $ctrs --output-format csv -d "$ROOT/$out/pmc_$name" -o "$name" -- python3 "$ROOT/tools/bench_dense_blocksum.py" --reps 4 > "$ROOT/$out/pmc_$name.log" 2>&1 )
}
job_pmc_dense()   { pmc_dense dfetch "FETCH_SIZE" && pmc_dense dwrite "WRITE_SIZE" && python - "$out" <<'PY' > "$out/pmc_dense_summary.txt"
import csv, glob, sys
out = sys.argv[1]
for tag, ctr in (("dfetch", "FETCH_SIZE"), ("dwrite", "WRITE_SIZE")):
    vals = []
    for fn in glob.glob(f"{out}/pmc_{tag}/**/*counter_collection.csv", recursive=True):
        acc = {}
        for r in csv.DictReader(open(fn)):
            if "dense_blocksum_pairs" in r["Kernel_Name"] and r["Counter_Name"] == ctr:
                acc[r["Dispatch_Id"]] = acc.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
        vals += list(acc.values())
    if vals:
        v = sorted(vals)[len(vals) // 2]
        print(f"{ctr}: median {v:.0f} KiB per launch = {v * 1024 / 1e9:.3f} GB over {len(vals)} launches "
              f"(algorithmic: 1.056 GB of kernel values read once + 0.032 GB of block sums read and written; gfx950 tallies the "
              f"128-B requests of 16-B-per-lane streaming reads at 64 B, MI355X_MICROARCH.md: FETCH_SIZE x 2 for such reads)")
PY
cat "$out/pmc_dense_summary.txt"; }
# stall / issue counters of the block sums: one tolerant pass per group (a counter this part does not offer fails only its own pass)
job_pmc_stalls()  { ( cd /tmp && rocprofv3 --list-avail > "$ROOT/$out/rocprof_list_avail.txt" 2>&1 ); for g in "a:SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "b:SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" "c:SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "d:SQ_WAVES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "e:SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VALU SQ_INSTS_VALU_MFMA_MOPS_F64"; do pmc "stall_${g%%:*}" "${g#*:}" || note "pass stall_${g%%:*} failed (kept going)"; done; return 0; }
job_tests_epoch() { timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "epoch or compaction_of_several or one_chunk_per or column_epochs or descriptor_driven" > "$out/gpu_tests_epoch.log" 2>&1; rc=$?; tail -15 "$out/gpu_tests_epoch.log" | cut -c1-300; return $rc; }
# block-sum micro-benchmark over tuning-variant builds: AB_LIBS="libbasq_hip.so libbasq_hip_w4pf1.so ..." (A B C A B C)
job_ab_bs()       { for i in 1 2 3; do for lib in ${AB_LIBS:-libbasq_hip.so}; do echo "== $lib" >> "$out/ab_bs.txt"; BASQ_HIP_LIB=$PWD/basq_amd/csrc/$lib timeout -k 10 120 python tools/bench_blocksum.py --reps 6 2>&1 | grep -v amdgpu.ids >> "$out/ab_bs.txt" || return 1; done; done; cat "$out/ab_bs.txt"; }
job_ab_columns()  { timeout -k 10 400 python tools/ab_engine.py IRR_COLUMNS 0 1 --reps 10 --visits 3 2>&1 | grep -v amdgpu.ids > "$out/ab_irr_columns.txt"; rc=$?; cat "$out/ab_irr_columns.txt"; return $rc; }
job_ab_columns_many() { ( for v in 0 1 0 1; do echo "== IRR_COLUMNS=$v"; timeout -k 10 200 python tools/bench_many.py --batches 24 --inflight 2,3,4 --set IRR_COLUMNS=$v 2>&1 | grep -v amdgpu.ids; done; for c in cfg2_rbf_1e5 rbf_2e4_defaults; do for v in 0 1 0 1; do echo "== $c IRR_COLUMNS=$v"; timeout -k 10 200 python tools/bench_many.py --case $c --batches 32 --inflight 2,4 --set IRR_COLUMNS=$v 2>&1 | grep -v amdgpu.ids; done; done ) > "$out/ab_irr_columns_many.txt" 2>&1; rc=$?; cut -c1-220 "$out/ab_irr_columns_many.txt"; return $rc; }
job_trace_round6() { prof round bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-concurrent --no-roofline-batch && python tools/trace_round.py "$(ls $out/prof_round/*kernel_trace.csv | head -1)" --first 1 --count 6 > "$out/trace_round.txt" 2>&1; rc=$?; cat "$out/trace_round.txt" | cut -c1-140; python tools/trace_batch.py "$(ls $out/prof_round/*kernel_trace.csv | head -1)" > "$out/trace_batch.txt" 2>&1; tail -45 "$out/trace_batch.txt" | cut -c1-150; return $rc; }
job_ab_maxc()     { timeout -k 10 400 python tools/ab_engine.py MAX_CLASSES 16 32 64 --reps 8 --visits 3 2>&1 | grep -v amdgpu.ids > "$out/ab_max_classes.txt"; rc=$?; cat "$out/ab_max_classes.txt"; return $rc; }
job_ab_bs_matern() { for i in 1 2 3; do for lib in ${AB_LIBS:-libbasq_hip.so}; do echo "== $lib" >> "$out/ab_bs_matern.txt"; BASQ_HIP_LIB=$PWD/basq_amd/csrc/$lib timeout -k 10 120 python tools/bench_blocksum.py --family matern52 --d 32 --n 200 --reps 3 2>&1 | grep -v amdgpu.ids >> "$out/ab_bs_matern.txt" || return 1; done; done; cat "$out/ab_bs_matern.txt"; }
job_ab_side_many() { ( for v in 0 1 0 1; do echo "== PIPELINED_SIDE_STREAM=$v"; timeout -k 10 200 python tools/bench_many.py --batches 24 --inflight 2,3,4 --set PIPELINED_SIDE_STREAM=$v 2>&1 | grep -v amdgpu.ids; done; for c in cfg2_rbf_1e5 rbf_2e4_defaults; do for v in 0 1 0 1; do echo "== $c PIPELINED_SIDE_STREAM=$v"; timeout -k 10 200 python tools/bench_many.py --case $c --batches 32 --inflight 2,4 --set PIPELINED_SIDE_STREAM=$v 2>&1 | grep -v amdgpu.ids; done; done ) > "$out/ab_side_many.txt" 2>&1; rc=$?; grep -v "^config\|^case" "$out/ab_side_many.txt" | cut -c1-130; return $rc; }
job_ab_apply_xcd() { ( for sh in "100 200" "50 100"; do for v in 0 1 0 1; do echo "== $sh BASQ_NS_APPLY_XCD=$v"; BASQ_NS_APPLY_XCD=$v timeout -k 10 120 python tools/bench_reduction.py $sh --reps 200 2>&1 | grep -E "nullspace|car_eliminate|LAPACK"; done; done ) > "$out/ab_apply_xcd.txt" 2>&1; rc=$?; cat "$out/ab_apply_xcd.txt"; return $rc; }
job_ab_apply_pairs() { ( for sh in "100 200" "50 100" "200 400"; do for v in 0 1 0 1; do echo "== $sh BASQ_NS_APPLY_PAIRS=$v"; BASQ_NS_APPLY_PAIRS=$v timeout -k 10 120 python tools/bench_reduction.py $sh --reps 200 2>&1 | grep -E "nullspace|LAPACK"; done; done ) > "$out/ab_apply_pairs.txt" 2>&1; rc=$?; cat "$out/ab_apply_pairs.txt"; return $rc; }
job_ab_bs_dims()  { for i in 1 2; do for lib in ${AB_LIBS:-libbasq_hip.so}; do for cfg in "rbf 10 100" "rbf 24 100" "matern52 24 100" "matern52 6 100" "rbf 34 100"; do set -- $cfg; echo "== $lib $cfg" >> "$out/ab_bs_dims.txt"; BASQ_HIP_LIB=$PWD/basq_amd/csrc/$lib timeout -k 10 120 python tools/bench_blocksum.py --family $1 --d $2 --n $3 --reps 3 2>&1 | grep -v amdgpu.ids >> "$out/ab_bs_dims.txt" || return 1; done; done; done; cat "$out/ab_bs_dims.txt"; }
job_attribute()   { timeout -k 10 900 python tools/attribute_mismatch.py > "$out/attribute_mismatch.txt" 2>&1; rc=$?; grep -v "amdgpu.ids\|^\[{" "$out/attribute_mismatch.txt" | cut -c1-220; return $rc; }
job_pmc()         { pmc fetch "FETCH_SIZE" && pmc write "WRITE_SIZE" && pmc pipe "SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" && python tools/pmc_summary.py "$out" > "$out/pmc_summary.json"; rc=$?; cut -c1-400 "$out/pmc_summary.json"; return $rc; }

for j in "$@"; do
    note "job $j"
    if ! "job_$j"; then
        note "job $j FAILED (rc != 0): stopping"
        exit 1
    fi
done
note "all jobs done"
