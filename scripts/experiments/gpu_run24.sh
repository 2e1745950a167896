#!/bin/bash
# what the weights kernel of the 12-byte planes waits for: variants without the maxes loads (1), without their LDS reads (2),
# without the xs loads (4), without the pair stores (8); timing only (results are wrong by construction)
mkdir -p gpurun_out
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
for v in 1 2 4 8; do
  rm -rf /tmp/tr
  DIVANS_HIP_LIBRARY=$REPO/gpurun_exp/libdivans_mw$v.so timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o trace -- python $REPO/bench.py --config mixing --steps 2 --warmup 1 --no-cpu-baseline --no-verify --check-streams 0 > /tmp/b.json 2>/tmp/tr.log
  echo "variant $v: $(find /tmp/tr -name '*kernel_stats*' -exec grep -h 'mix_weights\|rans_encode2' {} \; | cut -d, -f1,2,4 | tr '\n' ' ')"
done | tee $REPO/gpurun_out/r24_weights_variants.txt
