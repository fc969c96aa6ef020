#!/bin/bash
# SQ / TCP / TCC counters of ONE kernel of one python tool, one rocprofv3 --pmc pass per counter group (never combined with a trace domain
# other than --kernel-trace; the program itself follows `--`).  On a GPU box from the repository root:
#   tools/pmc_kernel.sh r06_om_pmc orderwise_filter_om_kernel tools/filter_series_time.py     -> gpurun_out/r06_om_pmc.txt
tag=$1; kern=$2; shift 2
script=$GRAFT_REPO_ROOT/$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
groups=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY"
 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
 "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY"
 "SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"
 "TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum"
)
: > $out/$tag.txt
for g in "${groups[@]}"; do
  d=/tmp/pmc_${tag}_$(echo $g | cut -d' ' -f1)
  rm -rf $d
  rocprofv3 --pmc $g --kernel-trace --output-format csv -d $d -o run -- python3 $script "$@" > $d.log 2>&1
  echo "== $g" >> $out/$tag.txt
  python3 - "$d" "$kern" >> $out/$tag.txt <<'PY'
import csv, glob, collections, sys
files = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not files:
    print("   (no counter file: pass failed)")
    sys.exit(0)
agg = collections.defaultdict(list)
for r in csv.DictReader(open(files[0])):
    if sys.argv[2] in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(agg.items()):
    print("   %-32s per-dispatch %.4g (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
  rm -rf $d
done
cat $out/$tag.txt
