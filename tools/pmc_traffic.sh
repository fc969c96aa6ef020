#!/bin/bash
# HBM traffic of the synthesis kernel from rocprofv3 PMC counters (two separate passes: FETCH_SIZE and WRITE_SIZE do not
# fit one pass on gfx950, MI355X_MICROARCH.md "rocprofv3 PMC slots").  Writes gpurun_out/pmc_traffic_raw.txt.
out=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -o run -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --cov-parallels 0 --steps 3 --warmup 1 > $out/pmc_$c.log 2>&1
done
python3 - <<PY
import csv, glob, json, collections
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$out/pmc_%s/**/*counter_collection.csv" % c, recursive=True)[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "synthesis_fused" in r["Kernel_Name"] and r["Counter_Name"] == c]
    res[c] = {"per_dispatch_raw": sum(vals) / len(vals), "dispatches": len(vals)}
print(json.dumps(res))
open("$out/pmc_traffic_raw.json", "w").write(json.dumps(res))
PY
