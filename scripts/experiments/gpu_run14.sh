#!/bin/bash
# decode2 variants on the same box: product (async miss loads, 7 waves/SIMD attribute) vs compiler-placed waits and/or no attribute
mkdir -p gpurun_out
for V in product noasync now7 noasync_now7; do
  if [ $V = product ]; then unset DIVANS_HIP_LIBRARY; else export DIVANS_HIP_LIBRARY=$PWD/gpurun_exp/libdivans_$V.so; fi
  echo "=== variant $V"
  timeout 300 python scripts/decode2_sweep.py --config simple --reps 3 --geoms 32:0:0:0:31:5:5:5:7:2,32:0:0:0:31:5:5:5:7:3,32:0:0:0:31:5:5:5:8:3 2>&1 | grep -v amdgpu.ids
  timeout 300 python scripts/decode2_sweep.py --config mixing --streams 32768 --reps 3 --geoms 16:16:0:0:5:5:5:5:7:2,16:16:0:0:5:5:5:5:7:3 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r14_variants.log 2>&1
cat gpurun_out/r14_variants.log
