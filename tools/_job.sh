set -u
out=gpurun_out/r7z; mkdir -p $out
for i in 1 2; do
  for tree in . _r4; do
    echo "== tree $tree"
    ( cd $tree && timeout -k 10 200 python tools/bench_many.py --case rbf_2e4_defaults --batches 48 --inflight 2,4,6 2>&1 | grep -v "amdgpu.ids\|^case" )
  done
done | tee $out/ab_r4_vs_r5_small_in_flight.txt
