#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats of the two-model encoder pass with launch sequences of 32 768 (default) and 65 536 streams
# (does a second Weights wave per SIMD pay?).  Output: gpurun_out/r06b_mixing_kernel_stats_bucket_batch_<n>.csv
set -u
REPO=$(pwd)
mkdir -p $REPO/gpurun_out /tmp/divans_cache
cd /tmp && export TMPDIR=/tmp
for bb in 32768 65536; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$bb -o trace -- python $REPO/bench.py --config mixing --steps 2 --warmup 1 --no-cpu-baseline --host-data --input-cache /tmp/divans_cache --check-streams 64 --table-candidates 1 --bucket-batch $bb > $REPO/gpurun_out/r06b_trace_$bb.json 2> $REPO/gpurun_out/r06b_trace_$bb.log
  find /tmp/tr_$bb -name '*kernel_stats*' -exec cp {} $REPO/gpurun_out/r06b_mixing_kernel_stats_bucket_batch_$bb.csv \;
done
