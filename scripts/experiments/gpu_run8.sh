#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r8_pytest.log 2>&1
tail -5 gpurun_out/r8_pytest.log
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/r8_bench.json 2> gpurun_out/r8_bench.err
tail -c 3000 gpurun_out/r8_bench.json; tail -5 gpurun_out/r8_bench.err
timeout 300 python scripts/batch_container_rate.py > gpurun_out/r8_batch_rate.txt 2>&1
cat gpurun_out/r8_batch_rate.txt
