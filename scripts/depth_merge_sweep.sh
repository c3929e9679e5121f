#!/bin/bash
# step rate of the headline workload against the shape of the elimination tree (FC_ND_DEPTH bisections fused FC_ND_MERGE at a time)
out=gpurun_out/depth_merge_sweep.txt
: > $out
for dm in "10 2" "9 3" "8 2" "8 4" "10 5" "9 2" "12 3" "12 4" "12 2" "11 2"; do
  set -- $dm
  line=$(FC_ND_DEPTH=$1 FC_ND_MERGE=$2 timeout -k 10 120 python bench.py --no-cpu-baseline --no-large-spmv --no-replicas --steps 1500 2>/dev/null | tail -1)
  echo "depth $1 merge $2: $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(round(d["value"]), "steps/s; launches/step", r["launches_per_step"], "factor_nnz", r["factor_nnz"], "residual", d["solve_rel_residual_pre_refine"])' 2>&1 | tail -1)" >> $out
done
cat $out
