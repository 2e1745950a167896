#!/bin/bash
# byte-rank table layout (DIVANS_D2_PERM): parity subset, then same-box A/B against the numeric layout
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_general_streams.py -m gpu -x -q > gpurun_out/r20_pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r20_pytest.txt
tail -3 gpurun_out/r20_pytest.txt
run () {  # label, library
  DIVANS_HIP_LIBRARY=$2 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --check-streams 64 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1 simple', d['value'], d['kernel_ms'], d['bit_exact'])
for k,v in d['configs'].items(): print('$1', k, v['value'], v['kernel_ms'], v['bit_exact'])"
}
{ run perm ""; run numeric gpurun_exp/libdivans_noperm.so; run perm ""; run numeric gpurun_exp/libdivans_noperm.so; } > gpurun_out/r20_perm_ab.txt 2>&1
cat gpurun_out/r20_perm_ab.txt
