#!/bin/bash
# Runs on the GPU box: divans_batch_compress / _decompress with their batches cut into other numbers of slices (gpurun_exp/libdivans_slices_d<n>_e<m>.so, built
# with -DDIVANS_BATCH_DECODE_SLICES=<n> -DDIVANS_BATCH_ENCODE_SLICES=<m>), 16 384 x 64 KiB.  Output: gpurun_out/r06_decode_slices_2.txt
OUT=gpurun_out/r06_decode_slices_2.txt
: > $OUT
for lib in "" $(ls gpurun_exp/libdivans_slices_*.so) ""; do
  echo "== library ${lib:-product}" >> $OUT
  DIVANS_HIP_LIBRARY=$lib python scripts/batch_container_rate.py 16384 0 x 0 0,2,0,2 2>/dev/null | python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('   mixing', d['dynamic_context_mixing'], 'compress', d['compress_MBps'], d['compress']['total_ms'], 'ms | decompress', d['decompress_MBps'], 'MB/s', d['decompress']['total_ms'], 'ms', {k:v for k,v in d['decompress_host_phases'].items() if v})
" >> $OUT
done
cat $OUT
