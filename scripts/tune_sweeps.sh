#!/bin/bash
# Tuning aid: bench.py's mean sweep launch time for a list of "FC_UP_SPLIT|FC_SWEEP_GEOM" settings.
run() {
  FC_UP_SPLIT=$1 FC_SWEEP_GEOM=$2 python bench.py --no-cpu-baseline --no-large-spmv --steps 1500 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$2', round(d['value']), round(d['batched_steps_per_s']), round(d['roofline']['mean_launch_us'],3), d['solve_rel_residual_pre_refine'])"
}
while read -r split geom; do run "$split" "$geom"; done
