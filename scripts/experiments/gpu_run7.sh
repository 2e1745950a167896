#!/bin/bash
# PMC passes of the decode kernels (generation 3 default) on configs[1] and configs[2]; GPU sharding test
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_sharding.py -x -q -m gpu > gpurun_out/r7_pytest.log 2>&1
tail -3 gpurun_out/r7_pytest.log
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
for CFG in simple mixing; do
  S=65536; [ $CFG = mixing ] && S=32768
  for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
    D=/tmp/pmc_$RANDOM
    timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $SET -d $D -o pmc -- python $REPO/bench.py --config $CFG --streams $S --steps 1 --warmup 0 --no-cpu-baseline --no-verify > /dev/null 2> $D.log
    python - $D $CFG <<'PY'
import csv, sys, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for p in glob.glob(sys.argv[1] + '/**/*counter_collection*.csv', recursive=True):
    for row in csv.DictReader(open(p)):
        kn = row.get("Kernel_Name", "?")
        if 'lit_decode' not in kn: continue
        k = (kn.split('(')[0][-34:], row["Counter_Name"])
        agg[k][0] += float(row["Counter_Value"]); agg[k][1] += 1
for (kn, cn), (v, n) in sorted(agg.items()):
    print(f"{sys.argv[2]:7s} {kn:34s} {cn:24s} avg={v / n:.5g} n={n}")
PY
  done
done > $REPO/gpurun_out/r7_pmc.txt 2>&1
cat $REPO/gpurun_out/r7_pmc.txt
