#!/bin/bash
# kernel split of the two-model pass with the 12-byte record planes
mkdir -p gpurun_out
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o trace -- python $REPO/bench.py --config mixing --steps 2 --warmup 1 --no-cpu-baseline --check-streams 64 > /tmp/b.json 2>/tmp/tr.log
find /tmp/tr -name '*kernel_stats*' -exec cp {} $REPO/gpurun_out/r23_mixing_kernel_stats.csv \;
cut -c1-140 $REPO/gpurun_out/r23_mixing_kernel_stats.csv | head -14
grep -o '"value": [0-9.]*\|"kernel_ms": {[^}]*}' /tmp/b.json | head -4
