timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ffi.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*\|"kernel_ms": {[^}]*}\|"bit_exact": [a-z]*'
