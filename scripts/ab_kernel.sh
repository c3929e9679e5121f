#!/bin/bash
# A/B of single kernels inside the default step: average duration by rocprofv3 for every value of an environment switch.
#   scripts/ab_kernel.sh FC_GATHER_ELL "0 1" "fc_rhs_gather|fc_early"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
VAR=$1; VALS=$2; PAT=$3
for v in $VALS $VALS; do
  export $VAR=$v
  rm -rf gpurun_out/ab_prof
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_prof -- python bench.py --steps 1500 --warmup 50 --no-cpu-baseline --no-large-spmv --no-replicas --no-other-configs > gpurun_out/ab_bench.json 2> gpurun_out/ab.err
  f=$(find gpurun_out/ab_prof -name "*_kernel_stats.csv" | head -1)
  echo "$VAR=$v"
  grep -E "$PAT" "$f" | awk -F'",' '{split($1,a,"("); n=split($2,b,","); print "   ", substr(a[1],2,40), "calls", b[1], "avg_ns", b[3]}'
done
