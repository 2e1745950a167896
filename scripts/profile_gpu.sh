#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats of the bench command plus PMC passes (separate runs,
# --pmc never combined with trace domains other than kernel-trace).  Outputs under gpurun_out/prof_<tag>/.
set -u
TAG=${1:-r01}
shift || true
BENCH_ARGS=${@:-"--streams 32768 --steps 2 --warmup 1 --no-cpu-baseline --no-verify"}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "== kernel trace: python bench.py $BENCH_ARGS" | tee $OUT/cmd.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $REPO/bench.py $BENCH_ARGS > $OUT/bench_traced.json 2> $OUT/trace.log
find $OUT/trace -name '*kernel_stats*' -exec cp {} $OUT/kernel_stats.csv \;
pmc_pass () {
  local name=$1; shift
  timeout 240 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $OUT/pmc_$name -o pmc -- python $REPO/bench.py $BENCH_ARGS > /dev/null 2> $OUT/pmc_$name.log
  find $OUT/pmc_$name -name '*counter_collection*' -exec cp {} $OUT/pmc_$name.csv \;
}
if [ "${PMC:-1}" = "1" ]; then
pmc_pass fetch FETCH_SIZE
pmc_pass write WRITE_SIZE
pmc_pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pmc_pass ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
pmc_pass sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM
pmc_pass sq2 SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
fi
python - "$OUT" <<'PY'
import csv, sys, collections, os
out = sys.argv[1]
summary = []
ks = os.path.join(out, "kernel_stats.csv")
if os.path.exists(ks):
    summary.append("== kernel stats (rocprofv3 --kernel-trace --stats)")
    summary.extend(open(ks).read().splitlines()[:12])
for name in ("fetch", "write", "tcc", "ea", "sq", "sq2"):
    p = os.path.join(out, f"pmc_{name}.csv")
    if not os.path.exists(p):
        summary.append(f"== pmc {name}: missing"); continue
    agg = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(p)):
        k = (row.get("Kernel_Name", "?")[:60], row.get("Counter_Name", "?"))
        agg[k][0] += float(row.get("Counter_Value", 0)); agg[k][1] += 1
    summary.append(f"== pmc {name}: per-dispatch average (sum over dispatches / dispatches)")
    for (kn, cn), (v, n) in sorted(agg.items()):
        summary.append(f"{kn:60s} {cn:22s} avg={v / n:.6g} n={n}")
    os.remove(p) if os.path.getsize(p) > 2_000_000 else None
open(os.path.join(out, "summary.txt"), "w").write("\n".join(summary) + "\n")
print("\n".join(summary))
PY
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_tcc $OUT/pmc_ea $OUT/pmc_sq $OUT/pmc_sq2
