#!/bin/bash
# rans_encode2_kernel (one lane per chunk, one-wave workgroups) with the waves per SIMD pinned: 2 for configs[1] (2048 waves), 1 for a 32 768-stream sequence of configs[2]
mkdir -p gpurun_out
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
run () { # label lib cfg
  rm -rf /tmp/tr
  DIVANS_HIP_LIBRARY=$2 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o trace -- python $REPO/bench.py --config $3 --steps 2 --warmup 1 --no-cpu-baseline --check-streams 64 > /tmp/b.json 2>/tmp/tr.log
  echo "$1 $3: $(grep -o '"bit_exact": [a-z]*\|"encode_rans_pass": [0-9.]*' /tmp/b.json | head -2 | tr '\n' ' ') $(find /tmp/tr -name '*kernel_stats*' -exec grep -h 'rans_encode2' {} \; | cut -d, -f2,4)"
}
{ run default "" simple; run eu2 $REPO/gpurun_exp/libdivans_rans2.so simple; run default "" mixing; run eu1 $REPO/gpurun_exp/libdivans_rans1.so mixing; run eu2 $REPO/gpurun_exp/libdivans_rans2.so mixing; } | tee $REPO/gpurun_out/r27_rans_eu.txt
