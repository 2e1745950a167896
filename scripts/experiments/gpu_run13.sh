#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r13_pytest.log 2>&1
tail -3 gpurun_out/r13_pytest.log
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/r13_bench.json 2> gpurun_out/r13_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r13_bench.json').read())
print({k:d[k] for k in ('value','ms_per_step','kernel_ms','bit_exact')})
for k,v in d['configs'].items(): print(k, {x:v[x] for x in ('value','ms_per_step','kernel_ms','bit_exact') if x in v})
PY
bash scripts/profile_round3.sh r03b > gpurun_out/r13_profile.log 2>&1
tail -5 gpurun_out/r13_profile.log
timeout 300 python bench.py --steps 3 --warmup 1 --total-streams 16384 --no-cpu-baseline --config all > gpurun_out/r13_bench_strong16384.json 2> gpurun_out/r13_bench_strong.err
tail -c 1500 gpurun_out/r13_bench_strong16384.json
