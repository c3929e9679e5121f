#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel stats + HBM traffic counters for bench.py.
# Counters go in their own passes (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950), with few
# steps (counter collection serialises every kernel).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profile
rm -rf "$OUT" && mkdir -p "$OUT"
(while true; do sleep 60; echo "[$(date +%T)] still profiling"; done) &
HB=$!
trap 'kill $HB 2>/dev/null' EXIT
# one stream, one record (FC_OVERLAP_TAIL=0): a clean launch sequence per step for the per-position medians; the kernels are the same
export FC_OVERLAP_TAIL=0
ARGS="--warmup 20 --no-cpu-baseline --no-large-spmv --no-replicas --no-other-configs --no-krylov"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python bench.py --steps ${STEPS:-1000} $ARGS > "$OUT/bench_stats.json" 2> "$OUT/stats.err"
echo "stats pass done"
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python bench.py --steps 100 $ARGS > "$OUT/bench_fetch.json" 2> "$OUT/fetch.err"
echo "fetch pass done"
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python bench.py --steps 100 $ARGS > "$OUT/bench_write.json" 2> "$OUT/write.err"
echo "write pass done"
python scripts/summarize_profile.py "$OUT" > "$OUT/summary.txt" 2>&1 || true
python - <<PY || true
import json, subprocess
p = "$OUT/traffic.json"
d = json.load(open(p))
d["commit"] = "${FC_COMMIT:-unknown}"
json.dump(d, open(p, "w"), indent=1)
PY
rm -rf "$OUT/stats" "$OUT/pmc_fetch" "$OUT/pmc_write"
tail -45 "$OUT/summary.txt"
