set -u
out=gpurun_out/r7c4; mkdir -p $out
ROOT=$(pwd)
( cd /tmp && TMPDIR=/tmp timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$out/prof" -o c4 -- python3 "$ROOT/tools/bench_configs.py" --only cfg4_matern52_1e6_d32 --reps 4 > "$ROOT/$out/prof.log" 2>&1 )
grep cfg4 $out/prof.log | cut -c1-250
head -12 $out/prof/c4_kernel_stats.csv | cut -c1-150
for sh in "200 400" "150 300"; do timeout -k 10 120 python tools/bench_reduction.py $sh --reps 50 2>&1 | grep -E "nullspace|car_el" | sed "s/^/[$sh] /"; done
