#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch_containers.py tests/test_gpu_general_streams.py -x -q -m gpu -k "decoder or generations or generic or brotli or batch or segment or general or ir_" 2>&1 | tail -3
timeout 600 python scripts/decode2_sweep.py --config mixing --streams 32768 --geoms 16:16:0:0:5:5:5:5:7:3,16:16:0:0:5:5:5:5:7:2,16:16:0:0:5:5:5:5:6:3,32:16:0:0:5:5:5:5:5:3,16:16:0:16:5:5:5:6:5:3,16:8:0:0:5:5:5:5:8:3 > gpurun_out/r11_sweep_mixing.log 2>&1
cat gpurun_out/r11_sweep_mixing.log
timeout 300 python scripts/batch_container_rate.py 4096 > gpurun_out/r11_batch_rate_4096.txt 2>&1; cat gpurun_out/r11_batch_rate_4096.txt
timeout 400 python scripts/batch_container_rate.py 16384 > gpurun_out/r11_batch_rate_16384.txt 2>&1; cat gpurun_out/r11_batch_rate_16384.txt
