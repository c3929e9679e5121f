#!/bin/bash
# A/B of the up-sweep's form (row form = product path; FC_UP_FORM=column: block + fold launches per level) on configs 4 / 5 / 3 and O1
cd "$GRAFT_REPO_ROOT"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["steps_per_s"] if "steps_per_s" in d else d["value"],1), d.get("worst_relative_residual"), d["y_last"][:2])'
for rep in 1 2; do
for c in refined1 pinball cavity_fine; do
  n=600; [ $c = cavity_fine ] && n=300
  for f in row column; do
    echo "$c $f: $(FC_UP_FORM=$f python scripts/bench_case.py $c --steps $n 2>/dev/null | python -c "$P")"
  done
done
for f in row column; do
  echo "O1 $f: $(FC_UP_FORM=$f python bench.py --steps 1500 --warmup 50 --no-cpu-baseline --no-large-spmv --no-replicas --no-other-configs 2>/dev/null | python -c "$P")"
done
done
