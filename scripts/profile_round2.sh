#!/bin/bash
# Runs on the GPU box (via gpurun): per BASELINE configuration, the kernel-trace stats of `python bench.py --config X` and the
# PMC passes that feed roofline.traffic (separate runs; --pmc is never combined with trace domains other than kernel-trace).
# Outputs under gpurun_out/prof_<tag>/<config>/.  usage: profile_round2.sh <tag>
set -u
TAG=${1:-r02}
REPO=$(pwd)
mkdir -p /tmp/divans_cache
cd /tmp && export TMPDIR=/tmp
run_config () {
  local CFG=$1; shift
  local FULL=$1; shift
  local ARGS="--config $CFG --steps 2 --warmup 1 --no-cpu-baseline --host-data --input-cache /tmp/divans_cache --check-streams 64 --table-candidates 1"
  local OUT=$REPO/gpurun_out/prof_$TAG/$CFG
  mkdir -p $OUT
  echo "python bench.py $ARGS" > $OUT/cmd.txt
  if [ "${SKIP_TRACE:-0}" != "1" ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $REPO/bench.py $ARGS > $OUT/bench_traced.json 2> $OUT/trace.log
  find $OUT/trace -name '*kernel_stats*' -exec cp {} $OUT/kernel_stats.csv \;
  fi
  pmc_pass () {
    local name=$1; shift
    timeout 400 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $OUT/pmc_$name -o pmc -- python $REPO/bench.py $ARGS > /dev/null 2> $OUT/pmc_$name.log
    find $OUT/pmc_$name -name '*counter_collection*' -exec cp {} $OUT/pmc_$name.csv \;
  }
  pmc_pass fetch FETCH_SIZE
  pmc_pass write WRITE_SIZE
  if [ "$FULL" = "1" ]; then
    pmc_pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
    pmc_pass ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
    pmc_pass sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM
    pmc_pass sq2 SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
  fi
  python - "$OUT" <<'PY'
import csv, sys, collections, os
out = sys.argv[1]
summary = []
ks = os.path.join(out, "kernel_stats.csv")
if os.path.exists(ks):
    summary.append("== kernel stats (rocprofv3 --kernel-trace --stats)")
    summary.extend(open(ks).read().splitlines()[:14])
for name in ("fetch", "write", "tcc", "ea", "sq", "sq2"):
    p = os.path.join(out, f"pmc_{name}.csv")
    if not os.path.exists(p):
        continue
    agg = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(p)):
        if "divans" not in row.get("Kernel_Name", ""):
            continue
        k = (row.get("Kernel_Name", "?")[:60], row.get("Counter_Name", "?"))
        agg[k][0] += float(row.get("Counter_Value", 0)); agg[k][1] += 1
    summary.append(f"== pmc {name}: per-dispatch average (sum over dispatches / dispatches)")
    for (kn, cn), (v, n) in sorted(agg.items()):
        summary.append(f"{kn:60s} {cn:22s} avg={v / n:.6g} n={n}")
    os.remove(p)
open(os.path.join(out, "summary.txt"), "w").write("\n".join(summary) + "\n")
print("\n".join(summary[:40]))
PY
  rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_tcc $OUT/pmc_ea $OUT/pmc_sq $OUT/pmc_sq2
}
run_config simple 1
run_config mixing 1
run_config decode_only ${PROFILE_FULL_ALL:-0}
if [ "${PROFILE_BINARY:-0}" = "1" ]; then run_config simple_binary ${PROFILE_FULL_ALL:-0}; fi
if [ "${SKIP_BENCH:-0}" != "1" ]; then
cd $REPO && python bench.py > gpurun_out/prof_$TAG/bench_line.json 2> gpurun_out/prof_$TAG/bench_line.err
tail -c 600 gpurun_out/prof_$TAG/bench_line.json
fi
