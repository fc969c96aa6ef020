#!/bin/bash
# SQ / TCP / TCC counters of the two headline kernels, one rocprofv3 --pmc pass per counter group (never combined with a trace
# domain other than --kernel-trace).  Run on a GPU box from the repository root:
#   tools/pmc_summary.sh r05        -> gpurun_out/r05_synthesis_pmc.txt, r05_covprop_pmc.txt, r05_filters_block_pmc.txt, r05_filters_dense_pmc.txt
tag=${1:-r06}
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
groups=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY"
 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
 "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY"
 "TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES TCP_GATE_EN1 TCP_TA_TCP_STATE_READ"
 "TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum"
)
run_group() {   # tag kernel-substring command...
  local tag=$1 kern=$2; shift 2
  : > $out/$tag.txt
  for g in "${groups[@]}"; do
    d=$out/pmc_${tag}_$(echo $g | cut -d' ' -f1)
    rm -rf $d
    rocprofv3 --pmc $g --kernel-trace --output-format csv -d $d -o run -- "$@" > $d.log 2>&1
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
  done
}
run_group ${tag}_synthesis_pmc synthesis_rot_kernel python3 $GRAFT_REPO_ROOT/bench.py --legs synthesis --cpu-sample 0 --steps 20 --warmup 5 --ramp 50 --idle-pass 0 --api-chain 0 --stage-limit-pass 0
run_group ${tag}_covprop_pmc gemm_f64_kernel python3 $GRAFT_REPO_ROOT/tools/gemm_phases.py --release-library 24
run_group ${tag}_filters_block_pmc orderwise_filter_om_kernel python3 $GRAFT_REPO_ROOT/bench.py --legs filters --cpu-sample 0 --steps 10 --warmup 2 --ramp 0 --idle-pass 0 --api-chain 0 --stage-limit-pass 0
run_group ${tag}_analysis_transform_pmc analysis_transform_kernel python3 $GRAFT_REPO_ROOT/bench.py --legs analysis --cpu-sample 0 --steps 10 --warmup 2 --ramp 0 --idle-pass 0 --api-chain 0 --stage-limit-pass 0
run_group ${tag}_analysis_operator_pmc analysis_operator_parity_kernel python3 $GRAFT_REPO_ROOT/bench.py --legs analysis --cpu-sample 0 --steps 10 --warmup 2 --ramp 0 --idle-pass 0 --api-chain 0 --stage-limit-pass 0
cp $out/${tag}_filters_block_pmc.txt $out/${tag}_filters_tmp.txt
run_group ${tag}_filters_dense_pmc gemm_tall_kernel python3 $GRAFT_REPO_ROOT/bench.py --legs filters --cpu-sample 0 --steps 10 --warmup 2 --ramp 0 --idle-pass 0 --api-chain 0 --stage-limit-pass 0
rm -f $out/${tag}_filters_tmp.txt
