#!/bin/bash
# usage: tools/pmc_any.sh <tag> "<counters>" <kernel-substring> <python script> -- PMC pass over an arbitrary script
tag=$1; counters=$2; kern=$3; script=$4
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $counters --kernel-trace --output-format csv -d $out -o run -- python3 $GRAFT_REPO_ROOT/$script > $out.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$out/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "$kern" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in agg.items():
    print("   %-32s per-dispatch %.4g (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
