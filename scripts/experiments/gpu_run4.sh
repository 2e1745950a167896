#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "decoder2 or generations" > gpurun_out/r4_pytest.log 2>&1
tail -4 gpurun_out/r4_pytest.log
timeout 600 python scripts/decode2_sweep.py --config simple --geoms 32:0:0:0:31:5:5:5:7:2,32:0:0:0:31:5:5:5:8:2,32:0:0:0:31:5:5:5:6:2,32:0:0:0:31:5:5:5:7:3,32:0:0:0:31:5:5:5:8:3,32:0:16:0:31:5:5:5:5:3,32:0:16:0:31:5:5:5:5:2,64:0:0:0:31:5:5:5:4:3 > gpurun_out/r4_sweep_simple.log 2>&1
cat gpurun_out/r4_sweep_simple.log
timeout 600 python scripts/decode2_sweep.py --config mixing --streams 32768 --geoms 16:16:0:0:5:5:5:5:7:2,16:16:0:0:5:5:5:5:6:2,16:16:0:0:5:5:5:5:7:3,16:16:0:0:5:5:5:5:6:3,32:16:0:0:5:5:5:5:5:3,16:16:0:16:5:5:5:6:5:3,16:16:0:16:5:5:5:6:5:2 > gpurun_out/r4_sweep_mixing.log 2>&1
cat gpurun_out/r4_sweep_mixing.log
