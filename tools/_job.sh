set -u
out=gpurun_out/r5n; mkdir -p $out
for i in 1 2; do for lib in libbasq_hip.so libbasq_hip_sqcopy.so; do echo "== $lib"; BASQ_HIP_LIB=$PWD/basq_amd/csrc/$lib timeout -k 10 120 python tools/bench_blocksum_sq.py 2>&1 | grep -v amdgpu.ids; done; done | tee $out/sq_ab.txt
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "wsabim_descriptor" 2>&1 | tail -2
timeout -k 10 600 python tools/bench_configs.py --only cfg5m_wsabim_5e5 --reps 6 2>&1 | grep cfg5m | cut -c1-200
