#!/bin/bash
# tuning experiment: factor-apply time of a large case against the (lanes:sub) geometry of the first up-sweep stages (FC_SWEEP_GEOM)
CASE=${1:-cavity_fine}
out=gpurun_out/sweep_geom_$CASE.txt
: > $out
for g in ${GEOMS:-"0:0"}; do
  line=$(FC_SWEEP_GEOM=$g timeout -k 10 200 python scripts/bench_case.py $CASE --steps 200 2>/dev/null | tail -1)
  echo "$g: $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["roofline"]["apply_us"],1), "us apply;", round(d["value"],1), "steps/s; residual", d["worst_relative_residual"])' 2>&1 | tail -1)" >> $out
done
cat $out
