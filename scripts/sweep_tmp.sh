timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ffi.py -x -q -m gpu 2>&1 | tail -5
for ep in 2 1; do timeout 200 python bench.py --streams 32768 --steps 2 --warmup 1 --no-cpu-baseline --encode-path $ep 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('path $ep', round(d['value'],1), d.get('kernel_ms'), d.get('bit_exact'))
"; done
