#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel stats + HBM traffic counters for bench.py.
# Counters go in their own passes (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profile
rm -rf "$OUT" && mkdir -p "$OUT"
ARGS="--steps ${STEPS:-1000} --warmup 20 --no-cpu-baseline --no-large-spmv"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python bench.py $ARGS > "$OUT/bench_stats.json" 2> "$OUT/stats.err"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python bench.py $ARGS > "$OUT/bench_fetch.json" 2> "$OUT/fetch.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python bench.py $ARGS > "$OUT/bench_write.json" 2> "$OUT/write.err"
python scripts/summarize_profile.py "$OUT" > "$OUT/summary.txt" 2>&1 || true
tail -40 "$OUT/summary.txt"
