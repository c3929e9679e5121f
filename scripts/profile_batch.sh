#!/bin/bash
# per-position kernel medians of the batched step (k = 16 / 8) + the throughput probe: scripts/profile_batch.sh [outdir]
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=${1:-gpurun_out/batch_prof}
rm -rf "$OUT" && mkdir -p "$OUT"
for K in 16 8; do
  FC_BATCH_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_b$K" -- python scripts/batch_probe.py --skip-parity --ks $K --steps 200 > "$OUT/batch_probe_rocprof_k$K.log" 2>&1
  python scripts/summarize_batch_profile.py "$OUT/prof_b$K" "$OUT/batch${K}_positions.csv" > "$OUT/batch${K}_positions.txt"
  cp $(find "$OUT/prof_b$K" -name "*_kernel_stats.csv" | head -1) "$OUT/batch${K}_kernel_stats.csv"
  rm -rf "$OUT/prof_b$K"
done
python scripts/batch_probe.py --steps 400 > "$OUT/batch_probe_O1.log" 2>&1
tail -8 "$OUT/batch_probe_O1.log"
cat "$OUT/batch16_positions.txt"
