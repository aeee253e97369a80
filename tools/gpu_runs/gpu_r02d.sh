#!/bin/bash
# round-2 GPU check #4: sweep of the LDS block-sum form's tuning knobs (TJ rows/lane, ST sets/wave, candidate prefetch)
set -u
out=gpurun_out/r02d; mkdir -p $out /tmp/v
build() {  # tj st pf waves
python - <<PY
from basq_amd import _build
_build.build(force=True, verbose=False, defines={"BASQ_LDS_TJ": $1, "BASQ_LDS_ST": $2, "BASQ_LDS_PREFETCH": $3, "BASQ_LDS_WAVES": $4}, out="/tmp/v/lib_$1_$2_$3_$4.so")
PY
}
cfgs="2_4_0_2 2_4_1_2 2_8_0_2 2_8_1_2 3_2_1_2 3_4_0_2 3_4_1_2 4_2_0_2 4_2_1_2 4_4_0_2 4_4_1_2 2_8_0_3 2_4_0_3 1_8_0_3 1_8_1_4 1_16_0_3 2_2_1_3 3_2_0_2"
for c in $cfgs; do IFS=_ read tj st pf wv <<< "$c"; build $tj $st $pf $wv > $out/build_$c.log 2>&1 & 
  while [ $(jobs -r | wc -l) -ge 9 ]; do sleep 1; done
done
wait
for c in $cfgs; do
  echo "== TJ_ST_PF_WAVES=$c" >> $out/sweep.txt
  BASQ_HIP_LIB=/tmp/v/lib_$c.so timeout 200 python tools/bench_blocksum.py --reps 3 2>&1 | grep -E "^lds|rel diff" >> $out/sweep.txt
done
BASQ_HIP_LIB=/tmp/v/lib_2_4_0_2.so timeout 200 python tools/bench_blocksum.py --reps 3 2>&1 | grep -E "^mfma" >> $out/sweep.txt
cat $out/sweep.txt
