set -u
out=gpurun_out/r7r; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_many.py tests/test_two_ranks_one_gpu.py -x -q -m gpu > $out/tests.log 2>&1; echo "tests rc=$?"; tail -2 $out/tests.log | cut -c1-200
timeout -k 10 600 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
j=json.load(open('gpurun_out/r7r/bench.json'))
print(j['value'], j['ms_per_step'], j['value_concurrent2'], j['value_concurrent3'], [c['latency_ms_median'] for c in j['concurrent']], j['roofline']['frac'], j['roofline']['self_check']['ok'], j['roofline']['shader_clock_note'])
PY
