set -u
out=gpurun_out/r5b; mkdir -p $out
for i in 1 2 3; do timeout -k 10 300 python tools/stall_probe.py --steps 60 > $out/stall_paused_$i.txt 2>&1 || exit 1; grep -v amdgpu.ids $out/stall_paused_$i.txt | cut -c1-330 | head -8; done
timeout -k 10 300 python tools/stall_probe.py --steps 60 --pause-gc 0 > $out/stall_unpaused.txt 2>&1; grep -v amdgpu.ids $out/stall_unpaused.txt | cut -c1-330 | head -6
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "two_ranks or structured_differential" > $out/new_tests.log 2>&1; echo "new tests rc=$?"; tail -3 $out/new_tests.log | cut -c1-300
