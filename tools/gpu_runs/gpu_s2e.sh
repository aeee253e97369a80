#!/bin/bash
# session-2 check e: PMC passes on the two tall-skinny GEMM forms (effective clock, pipe busy, waits)
set -u
out=gpurun_out/s2e; mkdir -p $out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/$out/counters.txt 2>&1
for v in sk_old sk_dpp_jt2; do
  export BASQ_HIP_LIB=$R/tools/variants/$v.so
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/$out/p1_$v -o p1 -- python3 $R/tools/prof_skinny.py > $R/$out/p1_$v.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $R/$out/p2_$v -o p2 -- python3 $R/tools/prof_skinny.py > $R/$out/p2_$v.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM --output-format csv -d $R/$out/p3_$v -o p3 -- python3 $R/tools/prof_skinny.py > $R/$out/p3_$v.log 2>&1
done
cd $R
for f in $(find $out -name "*counter_collection.csv"); do echo $f; grep skinny $f | awk -F, '{print $9" "$(NF-3)" "$(NF-2)" "($(NF)-$(NF-1))}' | cut -c1-30,100-200 | sort | uniq -c | head -12; done
