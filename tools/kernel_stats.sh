#!/bin/bash
# rocprofv3 --kernel-trace --stats of one python tool (the program itself follows `--`); prints the head of the per-kernel summary and copies
# it to gpurun_out/<tag>_kernel_stats.csv.  On a GPU box from the repository root:   tools/kernel_stats.sh r06_om tools/filter_series_time.py [args]
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
script=$GRAFT_REPO_ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
d=/tmp/prof_$tag
rm -rf $d
rocprofv3 --kernel-trace --stats --output-format csv -d $d -o run -- python3 $script "$@" > $out/${tag}_stdout.log 2>&1
f=$(find $d -name '*kernel_stats.csv' | head -1)
if [ -n "$f" ]; then cp $f $out/${tag}_kernel_stats.csv; python3 -c "
import csv, sys
for k, r in enumerate(csv.DictReader(open('$f'))):
    if k < int('${HEAD:-8}'): print('%-60s calls %6s  avg %9.2f us  min %9.2f  max %9.2f' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
"; else echo "no kernel stats"; tail -5 $out/${tag}_stdout.log; fi
rm -rf $d
