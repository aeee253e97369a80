set -u
out=gpurun_out/r5o; mkdir -p $out
ROOT=$(pwd)
( cd /tmp && TMPDIR=/tmp timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$out/prof" -o b -- python3 "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --plain > "$ROOT/$out/prof.log" 2>&1 )
tail -1 $out/prof.log | cut -c1-300
python tools/trace_batch.py "$(ls $out/prof/*kernel_trace.csv | head -1)" --batch -3 > $out/trace_batch.txt 2>&1; tail -42 $out/trace_batch.txt | cut -c1-200
