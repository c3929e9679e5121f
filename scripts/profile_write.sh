#!/bin/bash
# WRITE_SIZE pass only (short): the counter pass is slow, keep the step count small and print progress
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profile_w
rm -rf "$OUT" && mkdir -p "$OUT"
(while true; do sleep 60; echo "[$(date +%T)] still profiling"; done) &
HB=$!
timeout -k 10 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-large-spmv > "$OUT/bench_write.json" 2> "$OUT/write.err"
echo "rc=$?"
kill $HB
ls -R "$OUT" | head
