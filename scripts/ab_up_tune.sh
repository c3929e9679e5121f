#!/bin/bash
cd "$GRAFT_REPO_ROOT"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["steps_per_s"],1))'
for c in refined1 cavity_fine; do
  n=600; [ $c = cavity_fine ] && n=300
  echo "$c row: $(FC_UP_FORM=row python scripts/bench_case.py $c --steps $n 2>/dev/null | python -c "$P")"
  for sh in 0 1 -1; do for rc in 16 32 64; do
    echo "$c column lpr_shift=$sh rc=$rc: $(FC_UP_FORM=column FC_UPC_LPR_SHIFT=$sh FC_UPC_RC=$rc python scripts/bench_case.py $c --steps $n 2>/dev/null | python -c "$P")"
  done; done
done
