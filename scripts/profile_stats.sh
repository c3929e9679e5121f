#!/bin/bash
# Quick per-launch timing of one step (kernel trace only, no counters): scripts/profile_stats.sh <tag>
# Environment knobs (FC_BLOCK_MAXWD, FC_FUSED_TAIL, ...) are inherited by the profiled program.
set -e
TAG=${1:-run}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT" && mkdir -p "$OUT"
ARGS="--steps ${STEPS:-400} --warmup 20 --no-cpu-baseline --no-large-spmv ${EXTRA:-}"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python bench.py $ARGS > "$OUT/bench_stats.json" 2> "$OUT/stats.err"
python scripts/summarize_profile.py "$OUT" > "$OUT/summary.txt" 2>&1 || true
rm -rf "$OUT/stats"
grep -A40 "position" "$OUT/summary.txt" | cut -c1-150
