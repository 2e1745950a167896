timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
for cfg in "16 4 64,0" "8 4 32,0" "8 2 64,0" "8 3 32,0"; do set -- $cfg; python bench.py --streams 32768 --steps 2 --warmup 1 --no-cpu-baseline --no-verify --lanes $1 --blocks-per-cu $2 --split-cache $3 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$cfg', round(d['value'],1), d.get('kernel_ms'))
"; done
