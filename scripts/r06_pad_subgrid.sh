#!/bin/bash
# Runs on the GPU box (via gpurun): are the decoders bound by VALU issue when a batch does NOT fill the persistent grid?  N extra four-cycle VALU instructions per
# decoded byte (scripts/build_variants.sh pad_valu) on configs[3] (20 480 streams, mixing decoder: 237 VALU per byte and wave) and on one GPU's share of
# configs[4] (16 384 streams, plain decoder: 102).  Output: gpurun_out/r06_pad_subgrid.txt
OUT=gpurun_out/r06_pad_subgrid.txt
: > $OUT
for lib in "" gpurun_exp/libdivans_pad10.so gpurun_exp/libdivans_pad26.so gpurun_exp/libdivans_pad52.so ""; do
  for cfg in "--config decode_only" "--config simple --total-streams 16384" "--config mixing --total-streams 16384"; do
    echo "== library ${lib:-product} $cfg" >> $OUT
    DIVANS_HIP_LIBRARY=$lib python bench.py $cfg --steps 3 --warmup 1 --no-cpu-baseline --table-candidates 1 --check-streams 64 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
recs=[d] if d.get('value') else []
recs+=list(d.get('configs',{}).values())
for r in recs:
    print('   ', r.get('value'), r.get('unit'), {k.split('::')[-1][:34]:v for k,v in r['kernel_ms'].items()}, 'replay', r['roofline'].get('replay_ms'))
" >> $OUT
  done
done
cat $OUT
