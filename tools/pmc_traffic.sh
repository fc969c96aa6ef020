#!/bin/bash
# HBM traffic of the synthesis kernel from rocprofv3 PMC counters (two separate passes: FETCH_SIZE and WRITE_SIZE do not
# fit one pass on gfx950, MI355X_MICROARCH.md "rocprofv3 PMC slots") and the kernel-trace statistics of the same command.
# Run on a GPU box from the repository root: writes gpurun_out/r03_kernel_stats.csv and gpurun_out/r03_pmc_traffic.json.
out=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
args="--legs synthesis --cpu-sample 0 --steps 20 --warmup 5 --ramp 300"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_r03 -o run -- python3 $GRAFT_REPO_ROOT/bench.py $args > $out/prof_r03.log 2>&1
cp $(find $out/prof_r03 -name "*kernel_stats.csv" | head -1) $out/r03_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -o run -- python3 $GRAFT_REPO_ROOT/bench.py $args > $out/pmc_$c.log 2>&1
done
python3 - <<PY
import csv, glob, json
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$out/pmc_%s/**/*counter_collection.csv" % c, recursive=True)[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "synthesis_rot_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c]
    res[c] = {"per_dispatch_raw": sum(vals) / len(vals), "dispatches": len(vals)}
fetch_kb, write_kb = res["FETCH_SIZE"]["per_dispatch_raw"], res["WRITE_SIZE"]["per_dispatch_raw"]
summary = {
    "kernel": "synthesis_rot_kernel (240 epochs, d/o 96 -> 0.25 deg)",
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in two separate passes (tools/pmc_traffic.sh), per dispatch",
    "FETCH_SIZE_KB_raw": fetch_kb, "WRITE_SIZE_KB_raw": write_kb, "dispatches": res["FETCH_SIZE"]["dispatches"],
    "write_bytes": write_kb * 1024.0, "fetch_bytes_corrected": fetch_kb * 1024.0 * 2.0,
    "correction": "MI355X_MICROARCH.md HBM section: counters are in KB; on gfx950 FETCH_SIZE reports half of the bytes of coalesced streaming reads (16 B per lane loads and LDS-DMA alike) -> doubled; WRITE_SIZE is exact for 16-byte-per-lane stores. The fetch figure includes Infinity-Cache hits.",
    "lon_stage_bytes_per_launch": write_kb * 1024.0 + fetch_kb * 1024.0 * 2.0,
    "algorithmic_bytes_per_launch": 240 * 8 * (97 * 97 + 720 * 1440),
}
open("$out/r03_pmc_traffic.json", "w").write(json.dumps(summary, indent=1))
print(json.dumps(summary))
PY
