#!/bin/bash
cd "$GRAFT_REPO_ROOT"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["steps_per_s"],1), "apply_us", round(d["roofline"]["apply_us"],1), "launches", d["roofline"]["launches_per_apply"], "GB", round(d["roofline"]["bytes_per_apply"]/1e9,3))'
run() { echo "$1 shape=$2: $(FC_ND_SHAPE=$2 python scripts/bench_case.py $1 --steps $3 2>/dev/null | python -c "$P")"; }
run refined1 2,2,2,2,2,2 600; run refined1 3,3,2,2,2 600; run refined1 3,2,2,2,3 600; run refined1 2,2,2,3,3 600; run refined1 3,3,3,3 600; run refined1 3,2,2,2,2,2 600; run refined1 2,2,2,2,2,2 600
run pinball 2,2,2,2,2,2 600; run pinball 3,3,2,2,2 600; run pinball 3,2,2,2,3 600
run cavity_fine 2,2,2,2,2,2,2 300; run cavity_fine 3,3,2,2,2,2 300; run cavity_fine 3,2,2,2,2,3 300; run cavity_fine 2,2,2,2,2,3,1 300
