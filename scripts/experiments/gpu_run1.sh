#!/bin/bash
# round-3 GPU batch 1: parity of the second-generation decoder, decode sweeps, extra issue-rate measurements
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "decoder or generations or lengths or generic or brotli or uniform or ragged or many_streams or adversarial" > gpurun_out/r1_pytest.log 2>&1
tail -5 gpurun_out/r1_pytest.log
timeout 600 python scripts/decode2_sweep.py --config simple > gpurun_out/r1_sweep_simple.log 2>&1
cat gpurun_out/r1_sweep_simple.log
timeout 600 python scripts/decode2_sweep.py --config mixing --streams 32768 > gpurun_out/r1_sweep_mixing.log 2>&1
cat gpurun_out/r1_sweep_mixing.log
timeout 300 scripts/ubench/issue_rates > gpurun_out/r1_issue_rates.txt 2>&1
grep -E "waves/SIMD 8|waves/SIMD 4 " gpurun_out/r1_issue_rates.txt | grep -E "64|ffb|pk_fma|chain|ds_|bperm" | cut -c1-110
