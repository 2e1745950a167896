python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_unit_tests.py tests/test_gpu_batch_containers.py -q -x -k "rans or division or chunk or container or selftest or nibble_helper or byte_order" > gpurun_out/r05c_tests.log 2>&1; tail -3 gpurun_out/r05c_tests.log
for lib in "" gpurun_exp/libdivans_ransold.so "" gpurun_exp/libdivans_ransold.so; do
  DIVANS_HIP_LIBRARY=$lib python bench.py --config simple --no-cpu-baseline --table-candidates 1 --steps 4 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('lib=$lib', d['value'], 'bit_exact', d['bit_exact'], d['kernel_ms'])"
done
python bench.py --total-streams 16384 --no-cpu-baseline > gpurun_out/r05c_bench_16384.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r05c_bench_16384.json')); print(d['value'], d['bit_exact'], d['kernel_ms']); [print(k, v['value'], v['bit_exact'], v['kernel_ms']) for k,v in d.get('configs',{}).items()]"
