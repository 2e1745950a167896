#!/bin/bash
# Runs on the GPU box: is it the runtime's number of hardware queues (GPU_MAX_HW_QUEUES, default 4) that keeps more than four of the batch decompressor's
# slices from running at once?  Eight slices (gpurun_exp/libdivans_slices_d8_e2.so) and four (product) under 4 and 8 queues.  Output: gpurun_out/r06_hw_queues.txt
OUT=gpurun_out/r06_hw_queues.txt
: > $OUT
for q in "" 8 2; do
  for lib in "" gpurun_exp/libdivans_slices_d8_e2.so; do
    echo "== GPU_MAX_HW_QUEUES=${q:-default}  library ${lib:-product (4 slices in flight)}" >> $OUT
    if [ -n "$q" ]; then export GPU_MAX_HW_QUEUES=$q; else unset GPU_MAX_HW_QUEUES; fi
    DIVANS_HIP_LIBRARY=$lib python scripts/batch_container_rate.py 16384 0 x 0 0,2,0 2>/dev/null | python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('   mixing', d['dynamic_context_mixing'], 'compress', d['compress_MBps'], '| decompress', d['decompress_MBps'], 'MB/s', d['decompress']['total_ms'], 'ms, wait_gpu', d['decompress_host_phases']['wait_gpu_ms'])
" >> $OUT
  done
done
cat $OUT
