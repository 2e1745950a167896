#!/bin/bash
mkdir -p gpurun_out
for G in 1 3 2 1 3; do
  timeout 300 python bench.py --config simple --steps 5 --warmup 2 --no-cpu-baseline --check-streams 64 --decoder-generation $G 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('generation $G', d['value'], d['ms_per_step'], d['kernel_ms'], d['bit_exact'])"
done > gpurun_out/r15_bench_generations.txt 2>&1
cat gpurun_out/r15_bench_generations.txt
