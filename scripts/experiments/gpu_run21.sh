#!/bin/bash
# round 3, last state: full GPU tier, smoke, default bench line
mkdir -p gpurun_out
timeout 700 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r21_pytest.txt
timeout 150 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/r21_smoke.txt
timeout 400 python bench.py > gpurun_out/r21_bench.json 2> gpurun_out/r21_bench.err; echo "bench rc $?"
cut -c1-400 gpurun_out/r21_bench.json
