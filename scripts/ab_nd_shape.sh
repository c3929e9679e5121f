#!/bin/bash
cd "$GRAFT_REPO_ROOT"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print(round(d["value"]), "apply_us", round(r["launches_per_step"]*r["mean_launch_us"],1), "launches", r["launches_per_step"], "MB", round(r["bytes_per_launch"]*r["launches_per_step"]/1e6,1), "frac", round(r["frac"],3))'
for sh in ${SHAPES:-"2,2,2,2,2" "3,3,2,2" "2,2,2,2,2" "3,3,2,2" "3,2,3,2" "3,2,2,3" "3,3,3,2" "3,3,2,3" "2,2,2,2,2" "3,3,2,2" "3,3,1,1,2" "3,3,2,1,1" "3,4,3" "2,2,2,2,2"}; do
  echo "FC_ND_SHAPE=$sh: $(FC_ND_SHAPE=$sh timeout -k 5 200 python bench.py --steps 1500 --warmup 50 --no-cpu-baseline --no-large-spmv --no-replicas --no-other-configs 2>/dev/null | python -c "$P")"
done
