# encode+decode rate vs batch size for BASELINE configs[1] and configs[2]; DESIGN.md section 5
for cfg in simple mixing; do
for n in 2048 4096 8192 16384 32768 65536; do timeout 300 python bench.py --config $cfg --streams $n --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$cfg streams', $n, 'MB/s', round(d['value'],1), 'enc', d['encode_MBps'], 'dec', d['decode_MBps'], d['kernel_ms'], d['bit_exact'])
"; done; done
