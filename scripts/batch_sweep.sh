# encode+decode rate vs batch size (BASELINE configs[1]) and the mixing configuration at full size; DESIGN.md section 5
for n in 2048 4096 8192 16384 32768 65536; do timeout 300 python bench.py --streams $n --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('simple streams', $n, 'MB/s', round(d['value'],1), 'enc', d['encode_MBps'], 'dec', d['decode_MBps'], d['kernel_ms'], d['bit_exact'])
"; done
timeout 600 python bench.py --config mixing --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('mixing streams 65536 MB/s', round(d['value'],1), 'enc', d['encode_MBps'], 'dec', d['decode_MBps'], d['kernel_ms'], d['bit_exact'])
"
