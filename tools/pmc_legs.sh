#!/bin/bash
# HBM-side traffic of the kernels of every bench leg from rocprofv3 PMC counters: FETCH_SIZE and WRITE_SIZE in two SEPARATE passes per
# leg (they do not fit one pass on gfx950, MI355X_MICROARCH.md "rocprofv3 PMC slots"), --kernel-trace only beside them.  The program
# itself follows `--`.  Run on a GPU box from the repository root:   tools/pmc_legs.sh r04   -> gpurun_out/r04_pmc_traffic.json
tag=${1:-r06}
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
pass() {   # leg, counter, bench arguments...
  leg=$1; c=$2; shift; shift
  d=/tmp/pmc_${tag}_${leg}_$c
  rm -rf $d
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -o run -- python3 $GRAFT_REPO_ROOT/bench.py "$@" > $out/${tag}_pmc_${leg}_$c.log 2>&1
  f=$(find $d -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then cp $f $out/${tag}_pmc_${leg}_$c.csv; echo "$leg $c: $(wc -l < $f) rows"; else echo "$leg $c: no counter file"; tail -3 $out/${tag}_pmc_${leg}_$c.log; fi
  rm -rf $d
}
for c in FETCH_SIZE WRITE_SIZE; do
  pass synthesis $c --legs synthesis --cpu-sample 0 --steps 10 --warmup 2 --ramp 0 --idle-pass 0 --api-chain 0 --stage-limit-pass 0
  pass analysis $c --legs analysis --cpu-sample 0 --steps 10 --warmup 2 --ramp 0 --idle-pass 0 --api-chain 0 --stage-limit-pass 0
  pass filters $c --legs filters --cpu-sample 0 --steps 10 --warmup 2 --ramp 0 --idle-pass 0 --api-chain 0 --stage-limit-pass 0
  pass covariance $c --legs covariance --cpu-sample 0 --steps 2 --warmup 1 --ramp 0 --idle-pass 0 --api-chain 0 --stage-limit-pass 0 --cov-repeats 1 --cov-extensions 0 --cov-parallels 8
  pass smoother $c --legs smoother --cpu-sample 0 --steps 2 --warmup 1 --ramp 0 --idle-pass 0 --api-chain 0 --stage-limit-pass 0 --smoother-epochs 64 --smoother-repeats 1
done
python3 $GRAFT_REPO_ROOT/tools/pmc_legs_summary.py $out $tag
