#!/bin/bash
# round 3, last state after the 12-byte record planes: container tests (two-model batch path), smoke, default bench line
mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_gpu_batch_containers.py tests/test_gpu_sharding.py -m gpu -x -q 2>&1 | tail -2 | tee gpurun_out/r26_pytest.txt
timeout 100 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/r26_smoke.txt
timeout 300 python bench.py > gpurun_out/r26_bench.json 2> gpurun_out/r26_bench.err; echo "bench rc $?"
python -c "
import json
d=json.load(open('gpurun_out/r26_bench.json'))
print(d['value'], d['ms_per_step'], d['kernel_ms'], d['bit_exact'])
for k,v in d['configs'].items(): print(k, v['value'], v['kernel_ms'], v['bit_exact'], v.get('encoder_work_bytes_per_input_byte'))"
