#!/bin/bash
# 12-byte records of the two-model pass (xs + maxes planes): parity, then the mixing bench line
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/r22_pytest.txt
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --check-streams 64 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('simple', d['value'], d['kernel_ms'], d['bit_exact'])
for k,v in d['configs'].items(): print(k, v['value'], v['kernel_ms'], v['bit_exact'], v.get('encoder_work_bytes_per_input_byte'))" | tee gpurun_out/r22_bench.txt
