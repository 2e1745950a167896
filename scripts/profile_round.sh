#!/bin/bash
# Runs on the GPU box (via gpurun): per BASELINE configuration, the kernel-trace stats of `python bench.py --config X` and the
# PMC passes that feed roofline.traffic and the instruction census (separate runs; --pmc is never combined with trace domains
# other than kernel-trace).  Outputs under gpurun_out/prof_<tag>/<config>/.  usage: profile_round.sh <tag>  (rounds 3 and 4)
set -u
TAG=${1:-r04}
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
mkdir -p /tmp/divans_cache
run_config () {
  local CFG=$1; shift
  local FULL=$1; shift
  # --table-candidates 1: the averages are over the timed launches only, on the placement the first allocation got
  local ARGS="--config $CFG --steps 2 --warmup 1 --no-cpu-baseline --host-data --input-cache /tmp/divans_cache --check-streams 64 --table-candidates 1"
  [ -n "${PROFILE_FULL:-}" ] && FULL=$PROFILE_FULL      # PROFILE_FULL=0: kernel stats + FETCH_SIZE / WRITE_SIZE only, for every configuration
  local OUT=$REPO/gpurun_out/prof_$TAG/$CFG
  mkdir -p $OUT
  echo "python bench.py $ARGS" > $OUT/cmd.txt
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $REPO/bench.py $ARGS > $OUT/bench_traced.json 2> $OUT/trace.log
  find $OUT/trace -name '*kernel_stats*' -exec cp {} $OUT/kernel_stats.csv \;
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
        k = (row.get("Kernel_Name", "?")[:70], row.get("Counter_Name", "?"))
        agg[k][0] += float(row.get("Counter_Value", 0)); agg[k][1] += 1
    summary.append(f"== pmc {name}: per-dispatch average (sum over dispatches / dispatches)")
    for (kn, cn), (v, n) in sorted(agg.items()):
        summary.append(f"{kn:70s} {cn:22s} avg={v / n:.6g} n={n}")
    os.remove(p)
open(os.path.join(out, "summary.txt"), "w").write("\n".join(summary) + "\n")
print("\n".join(summary[:30]))
PY
  rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_tcc $OUT/pmc_ea $OUT/pmc_sq $OUT/pmc_sq2
}
run_config simple 1
run_config mixing 1
run_config decode_only 0
