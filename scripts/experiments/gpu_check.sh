timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /tmp/bench.json 2>/dev/null
grep -o '"value": [0-9.]*\|"kernel_ms": {[^}]*}\|"bit_exact": [a-z]*' /tmp/bench.json
find /tmp/tr -name '*kernel_stats*' -exec head -9 {} \; | cut -c1-120
