#!/bin/bash
# Runs on the GPU box: the FETCH / WRITE / TCC_EA0 passes of `bench.py --config simple_binary` (configs[1]'s options on non-text input), so that this
# sub-record's roofline.traffic is a counter too.  Output: gpurun_out/prof_r06/simple_binary/summary.txt
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_r06/simple_binary
mkdir -p $OUT /tmp/divans_cache
cd /tmp && export TMPDIR=/tmp
ARGS="--config simple_binary --steps 2 --warmup 1 --no-cpu-baseline --host-data --check-streams 64 --table-candidates 1"
echo "python bench.py $ARGS" > $OUT/cmd.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $REPO/bench.py $ARGS > $OUT/bench_traced.json 2> $OUT/trace.log
find $OUT/trace -name '*kernel_stats*' -exec cp {} $OUT/kernel_stats.csv \;
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "ea TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  set -- $pass; name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $OUT/pmc_$name -o pmc -- python $REPO/bench.py $ARGS > /dev/null 2> $OUT/pmc_$name.log
  find $OUT/pmc_$name -name '*counter_collection*' -exec cp {} $OUT/pmc_$name.csv \;
done
python - "$OUT" <<'PY'
import csv, sys, collections, os
out = sys.argv[1]
summary = ["== kernel stats (rocprofv3 --kernel-trace --stats)"] + [l for l in open(os.path.join(out, "kernel_stats.csv")).read().splitlines() if "divans" in l or l.startswith('"Name"')]
for name in ("fetch", "write", "ea"):
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
print("\n".join(summary[:30]))
PY
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_ea
