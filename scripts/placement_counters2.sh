#!/bin/bash
# Counters per decode launch for SEVERAL placements of the tables in one process (scripts/decode_once.py --codecs K: K codecs alive, each
# decoding twice): which counters move with the decode time from one placement to the next?  One rocprofv3 --pmc pass per counter group.
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$REPO/gpurun_out/r04q}; mkdir -p $OUT; OUT=$(cd $OUT && pwd)
K=${2:-10}
cd /tmp && export TMPDIR=/tmp
pass () {
  name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $OUT/$name -o pmc -- python $REPO/scripts/decode_once.py --config simple --streams 28672 --reps 2 --codecs $K > $OUT/$name.log 2>&1
  find $OUT/$name -name '*counter_collection*' -exec cp {} $OUT/$name.csv \;
  rm -rf $OUT/$name
}
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_sum
pass lat TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum
pass dram TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum
python3 - $OUT <<'PY'
import csv, glob, os, re, sys
out = sys.argv[1]
lines = []
for f in sorted(glob.glob(os.path.join(out, "*.csv"))):
    name = os.path.basename(f)[:-4]
    times = [float(x) for l in open(os.path.join(out, name + ".log")) if l.startswith("codec ") for x in re.findall(r"decode ([0-9. ]+) ms", l)[0].split()]
    per = {}
    for r in csv.DictReader(open(f)):
        if "lit_decode" not in r.get("Kernel_Name", ""):
            continue
        per.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(per)
    names = sorted({c for d in per.values() for c in d})
    lines.append(f"== pass {name}: one line per decode launch (launches 2k, 2k+1 = codec k); decode ms as the codec reported them under the profiler")
    lines.append("%8s " % "ms" + " ".join("%34s" % n for n in names))
    for i, d in enumerate(ids):
        lines.append("%8.2f " % (times[i] if i < len(times) else float("nan")) + " ".join("%34.5g" % per[d].get(n, float("nan")) for n in names))
open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
