#!/bin/bash
mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_gpu_batch_containers.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python scripts/batch_container_rate.py 4096 > gpurun_out/r10_batch_rate_4096.txt 2>&1; cat gpurun_out/r10_batch_rate_4096.txt
timeout 400 python scripts/batch_container_rate.py 16384 > gpurun_out/r10_batch_rate_16384.txt 2>&1; cat gpurun_out/r10_batch_rate_16384.txt
