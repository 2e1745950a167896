#!/bin/bash
# rans_encode2_kernel pinned to two waves per SIMD: parity subset + default bench line
mkdir -p gpurun_out
timeout 120 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_unit_tests.py -m gpu -x -q -k "rans or chunk or lengths or ragged or ans" 2>&1 | tail -2 | tee gpurun_out/r28_pytest.txt
timeout 200 python bench.py > gpurun_out/r28_bench.json 2> gpurun_out/r28_bench.err; echo "bench rc $?"
python -c "
import json
d=json.load(open('gpurun_out/r28_bench.json'))
print(d['value'], d['ms_per_step'], d['kernel_ms'], d['bit_exact'])
for k,v in d['configs'].items(): print(k, v['value'], v['kernel_ms'], v['bit_exact'])"
