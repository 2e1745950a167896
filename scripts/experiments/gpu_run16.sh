#!/bin/bash
# round 3: validation of the encoder work-memory diet (in-place unsort, rANS per sub-batch): GPU tests + default bench
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r16_pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r16_pytest.txt
tail -5 gpurun_out/r16_pytest.txt
timeout 900 python bench.py > gpurun_out/r16_bench.json 2> gpurun_out/r16_bench.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r16_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['kernel_ms'], d.get('encoder_work_bytes_per_input_byte'), d['bit_exact'])
for k, v in d.get('configs', {}).items():
    print(k, v.get('value'), v.get('kernel_ms'), v.get('encoder_work_bytes_per_input_byte'), v.get('bit_exact'))
PY
