set -u
out=gpurun_out/r5f; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "nullspace or fuzz_single or chol or cluster_red" > $out/ns_tests.log 2>&1; rc=$?; tail -2 $out/ns_tests.log | cut -c1-300; [ $rc -eq 0 ] || exit 1
for sh in "100 200" "50 100" "31 62" "200 400"; do timeout -k 10 120 python tools/bench_reduction.py $sh 2>&1 | grep -E "nullspace|car_el" | sed "s/^/[$sh] /"; done | tee $out/reduction.txt
timeout -k 10 120 tools/ns_prof 100 200 > $out/ns_prof.txt 2>&1; cat $out/ns_prof.txt | cut -c1-200
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "golden_parity and not cfg4 and not cfg5" > $out/golden.log 2>&1; echo "golden rc=$?"; tail -2 $out/golden.log | cut -c1-300
