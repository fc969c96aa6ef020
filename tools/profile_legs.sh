#!/bin/bash
# rocprofv3 --kernel-trace --stats of every leg of bench.py, one run per leg (the program itself follows `--`), summaries copied to
# gpurun_out/<tag>_<leg>_kernel_stats.csv.  Run on a GPU box from the repository root:   tools/profile_legs.sh r03
tag=${1:-r06}
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run_leg() {   # leg-name, bench arguments...
  leg=$1; shift
  d=/tmp/prof_${tag}_$leg
  rm -rf $d
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -o run -- python3 $GRAFT_REPO_ROOT/bench.py "$@" > $out/${tag}_${leg}_bench.json 2> $out/${tag}_${leg}_bench.err
  f=$(find $d -name '*kernel_stats.csv' | head -1)
  if [ -n "$f" ]; then cp $f $out/${tag}_${leg}_kernel_stats.csv; echo "$leg: $(wc -l < $f) kernels"; else echo "$leg: no kernel stats"; fi
  rm -rf $d
}
run_leg synthesis --legs synthesis --cpu-sample 0 --steps 20 --warmup 5 --ramp 300 --idle-pass 0 --api-chain 0 --stage-limit-pass 0
run_leg analysis --legs analysis --cpu-sample 0 --steps 20 --warmup 5 --ramp 0 --idle-pass 0 --api-chain 0 --stage-limit-pass 0
run_leg filters --legs filters --cpu-sample 0 --steps 20 --warmup 5 --ramp 0 --idle-pass 0 --api-chain 0 --stage-limit-pass 0
run_leg covariance --legs covariance --cpu-sample 0 --steps 5 --warmup 2 --ramp 0 --idle-pass 0 --api-chain 0 --stage-limit-pass 0 --cov-repeats 1 --cov-extensions 0
run_leg smoother --legs smoother --cpu-sample 0 --steps 5 --warmup 2 --ramp 0 --idle-pass 0 --api-chain 0 --stage-limit-pass 0 --smoother-epochs 256 --smoother-repeats 1
