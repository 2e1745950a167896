#!/bin/bash
# Runs on the GPU box (via gpurun): the counter calibration of VERDICT r04 item 2 and the VGPR-row-cache micro-benchmark of item 1b.
# Outputs under gpurun_out/r05_cal/.  Every --pmc pass is its own run with --kernel-trace only.
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/r05_cal
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B=$REPO/scripts/ubench/counter_calibration
timeout 120 $B > $OUT/known.csv 2> $OUT/known.err
pass () {
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $OUT/pmc_$name -o pmc -- $B > /dev/null 2> $OUT/pmc_$name.log
  find $OUT/pmc_$name -name '*counter_collection*' -exec cp {} $OUT/pmc_$name.csv \;
  rm -rf $OUT/pmc_$name
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
pass wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
python $REPO/scripts/ubench/calibration_table.py $OUT/known.csv $OUT/pmc_*.csv > $OUT/counter_calibration.txt 2>&1
cat $OUT/counter_calibration.txt
timeout 300 $REPO/scripts/ubench/vgpr_row_cache > $OUT/vgpr_row_cache.txt 2>&1
cat $OUT/vgpr_row_cache.txt
