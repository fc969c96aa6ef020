#!/bin/bash
# usage: tools/pmc_run.sh <tag> "<counters>" [bench args...]   -- one rocprofv3 --pmc pass over bench.py, kernel rows to gpurun_out/pmc_<tag>.csv
tag=$1; counters=$2; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $counters --kernel-trace --output-format csv -d $out -o run -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --steps 3 --warmup 1 "$@" > $out.log 2>&1
find $out -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $out.csv
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("$out.csv")))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
for k in agg:
    print(k)
    for c, v in agg[k].items():
        print("   %-32s total %.4g  per-dispatch %.4g  (n=%d)" % (c, v, v / cnt[(k, c)], cnt[(k, c)]))
PY
