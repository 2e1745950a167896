#!/bin/bash
# Counters of the decode kernel on a SLOW placement of the CDF tables (one physically contiguous block) and a fast one (2 MiB chunks),
# one rocprofv3 --pmc pass per counter group (never combined with trace domains other than --kernel-trace).
#   bash scripts/placement_counters.sh [out dir]      (on the GPU box; cd /tmp && export TMPDIR=/tmp first, as the guide says)
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$REPO/gpurun_out/r04p}; mkdir -p $OUT; OUT=$(cd $OUT && pwd)
cd /tmp && export TMPDIR=/tmp
pass () {   # mode, group name, counters...
  mode=$1; name=$2; shift 2
  DIVANS_TABLES_ALLOC=$mode timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $OUT/${mode//:/_}_$name -o pmc -- python $REPO/scripts/decode_once.py --config simple --streams 28672 --reps 2 > $OUT/${mode//:/_}_$name.log 2>&1
  find $OUT/${mode//:/_}_$name -name '*counter_collection*' -exec cp {} $OUT/${mode//:/_}_$name.csv \;
  rm -rf $OUT/${mode//:/_}_$name
}
for mode in contiguous scattered:2; do
  pass $mode utcl1 TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum
  pass $mode lat TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum
  pass $mode ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum
  pass $mode tcc TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum
  pass $mode misc GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY TCP_PENDING_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum
done
python3 - $OUT <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
rows = {}
for f in sorted(glob.glob(os.path.join(out, "*.csv"))):
    mode = os.path.basename(f).rsplit("_", 1)[0]
    for r in csv.DictReader(open(f)):
        if "lit_decode" not in r.get("Kernel_Name", ""):
            continue
        rows.setdefault((mode, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
names = sorted({k[1] for k in rows})
modes = sorted({k[0] for k in rows})
with open(os.path.join(out, "summary.txt"), "w") as fo:
    fo.write("%-50s %s\n" % ("counter (per decode launch, mean of the launches)", " ".join("%18s" % m for m in modes)))
    for n in names:
        fo.write("%-50s %s\n" % (n, " ".join("%18.4g" % (sum(rows[(m, n)]) / len(rows[(m, n)])) if (m, n) in rows else "%18s" % "-" for m in modes)))
    for f in sorted(glob.glob(os.path.join(out, "*.log"))):
        for line in open(f):
            if "decode" in line and "ms" in line:
                fo.write(os.path.basename(f) + ": " + line)
print(open(os.path.join(out, "summary.txt")).read())
PY
