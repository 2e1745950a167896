#!/bin/bash
# Runs on the GPU box (via gpurun): scripts/ubench/wb_policy under rocprofv3 --pmc, one pass per counter set.  Output: gpurun_out/r06_wb/.
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/r06_wb
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B=$REPO/scripts/ubench/wb_policy
timeout 120 $B > $OUT/known.csv 2> $OUT/known.err
pass () {
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $OUT/pmc_$name -o pmc -- $B > /dev/null 2> $OUT/pmc_$name.log
  find $OUT/pmc_$name -name '*counter_collection*' -exec cp {} $OUT/pmc_$name.csv \;
  find $OUT/pmc_$name -name '*kernel_trace*' -exec cp {} $OUT/trace_$name.csv \;
  rm -rf $OUT/pmc_$name
}
pass rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
pass wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass size FETCH_SIZE WRITE_SIZE
python $REPO/scripts/ubench/calibration_table.py $OUT/known.csv $OUT/pmc_*.csv > $OUT/wb_policy.txt 2>&1
cat $OUT/wb_policy.txt
