#!/bin/bash
# (a) parity of the two-model pass after the record-plane change; (b) chain kernels with at most one wave per SIMD enforced
# (amdgpu_waves_per_eu(1,1)), and with four context-model chain waves per CU on top
mkdir -p gpurun_out
REPO=$(pwd)
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mix or sub_batch or chunk" 2>&1 | tail -2 | tee gpurun_out/r25_pytest.txt
cd /tmp && export TMPDIR=/tmp
for v in default eu eu4; do
  for cfg in simple mixing; do
    [ "$v" = "eu4" ] && [ "$cfg" = "simple" ] && continue
    rm -rf /tmp/tr
    LIB=""; [ "$v" != "default" ] && LIB=$REPO/gpurun_exp/libdivans_$v.so
    DIVANS_HIP_LIBRARY=$LIB timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o trace -- python $REPO/bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --check-streams 64 > /tmp/b.json 2>/tmp/tr.log
    echo "$v $cfg: $(grep -o '"value": [0-9.]*\|"bit_exact": [a-z]*\|"encode_model_pass": [0-9.]*' /tmp/b.json | head -3 | tr '\n' ' ') $(find /tmp/tr -name '*kernel_stats*' -exec grep -h 'chain_kernel' {} \; | cut -d, -f1,4 | sed 's/divans_hip:://g; s/(divans_hip::[A-Za-z]*)//g' | tr '\n' ' ')"
  done
done | tee $REPO/gpurun_out/r25_chain_eu.txt
