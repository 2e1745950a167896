#!/bin/bash
# cache-policy bits on the decoder's row loads / stores (scripts/build_variants.sh cache_policy), both configs, same box
mkdir -p gpurun_out
run () {  # label, library
  for C in simple mixing; do
    DIVANS_HIP_LIBRARY=$2 timeout 300 python bench.py --config $C --steps 3 --warmup 1 --no-cpu-baseline --check-streams 64 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d if '$C'=='simple' else d['configs']['mixing']
print('$1 $C', r['value'], r['kernel_ms'], r['bit_exact'])"
  done
}
{
run product ""
for V in ldnt stnt ldstnt stsc1 ldsc0; do run $V gpurun_exp/libdivans_$V.so; done
run product ""
} > gpurun_out/r19_cache_policy.txt 2>&1
cat gpurun_out/r19_cache_policy.txt
