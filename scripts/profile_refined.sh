#!/bin/bash
# Per-launch-position kernel times of one step on the refined mesh (BASELINE config 4): scripts/profile_refined.sh
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_refined
rm -rf "$OUT" && mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python bench.py --refine ${REFINE:-1} --steps ${STEPS:-300} --warmup 20 --no-cpu-baseline --no-large-spmv --no-replicas > "$OUT/bench.json" 2> "$OUT/err.txt"
python scripts/summarize_profile.py "$OUT" > "$OUT/summary.txt" 2>&1 || true
rm -rf "$OUT/stats"
tail -40 "$OUT/summary.txt"
