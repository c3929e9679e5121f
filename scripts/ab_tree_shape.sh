#!/bin/bash
# tree shape against steps/s with the column-form up-sweep (two launches per up level shift the balance towards shallower trees?)
cd "$GRAFT_REPO_ROOT"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["steps_per_s"],1), "apply_us", round(d["roofline"]["apply_us"],1), "launches", d["roofline"]["launches_per_apply"], "GB", round(d["roofline"]["bytes_per_apply"]/1e9,3))'
run() { echo "$1 depth=$2 merge=$3: $(FC_ND_DEPTH=$2 FC_ND_MERGE=$3 python scripts/bench_case.py $1 --steps $4 2>/dev/null | python -c "$P")"; }
run refined1 12 2 600; run refined1 10 2 600; run refined1 12 3 600; run refined1 9 3 600; run refined1 12 4 600; run refined1 14 2 600
run cavity_fine 14 2 300; run cavity_fine 12 2 300; run cavity_fine 15 3 300; run cavity_fine 12 3 300; run cavity_fine 12 4 300; run cavity_fine 16 2 300
run pinball 12 2 600; run pinball 10 2 600; run pinball 12 3 600; run pinball 14 2 600
