#!/bin/bash
# usage: pmc_quick.sh "<bench args>" "CTR1 CTR2 ..." ["CTR.. second pass"] ...   -> prints per-kernel averages for divans kernels
set -u
REPO=$(pwd); ARGS=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "$@"; do
  i=$((i+1)); D=/tmp/pmcq_$i; rm -rf $D
  rocprofv3 --kernel-trace --output-format csv --pmc $SET -d $D -o pmc -- python $REPO/bench.py $ARGS > /dev/null 2> $D.log
  python - $D <<'PY'
import csv, sys, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for p in glob.glob(sys.argv[1] + '/**/*counter_collection*.csv', recursive=True):
    for row in csv.DictReader(open(p)):
        kn = row.get("Kernel_Name", "?")
        if 'divans' not in kn: continue
        k = (kn.split('(')[0][-40:], row["Counter_Name"])
        agg[k][0] += float(row["Counter_Value"]); agg[k][1] += 1
for (kn, cn), (v, n) in sorted(agg.items()):
    print(f"{kn:42s} {cn:26s} avg={v / n:.5g} n={n}")
PY
done
