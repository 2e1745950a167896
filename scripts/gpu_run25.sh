#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bucket or encode_paths or model_pass or sub_batches or speeds" > gpurun_out/r25_pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r25_pytest.txt
tail -5 gpurun_out/r25_pytest.txt
for i in 1 2; do timeout 300 python bench.py --config simple --steps 3 --warmup 1 --no-cpu-baseline --check-streams 64 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('simple', d['value'], d['kernel_ms'], d['bit_exact'])"; done
