#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "decoder2 or generations or generic or brotli or lengths" > gpurun_out/r3_pytest.log 2>&1
tail -4 gpurun_out/r3_pytest.log
timeout 600 python scripts/decode2_sweep.py --config simple --geoms 32:0:0:0:31:5:5:5:7,32:0:0:0:31:5:5:5:8,32:0:0:0:31:5:5:5:6,64:0:0:0:31:5:5:5:4,32:0:16:0:31:5:5:5:5,32:0:0:0:4:5:5:5:7 > gpurun_out/r3_sweep_simple.log 2>&1
cat gpurun_out/r3_sweep_simple.log
timeout 600 python scripts/decode2_sweep.py --config mixing --streams 32768 --geoms 16:16:0:0:5:5:5:5:7,16:16:0:0:5:5:5:5:6,32:16:0:0:5:5:5:5:5,16:8:0:0:5:5:5:5:8,16:16:0:16:5:5:5:5:5,16:16:0:0:8:5:5:5:7,16:16:0:0:3:31:5:5:7 > gpurun_out/r3_sweep_mixing.log 2>&1
cat gpurun_out/r3_sweep_mixing.log
