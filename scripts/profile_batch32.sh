set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/b32; rm -rf $OUT; mkdir -p $OUT
for K in ${KS:-32 16}; do
FC_OVERLAP_TAIL=0 FC_BATCH_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_b$K" -- python scripts/batch_probe.py --skip-parity --ks $K --steps 200 $PROBE_ARGS > "$OUT/probe_k$K.log" 2>&1
python scripts/summarize_batch_profile.py "$OUT/prof_b$K" "$OUT/batch${K}_positions.csv" > "$OUT/batch${K}_positions.txt"
rm -rf "$OUT/prof_b$K"
cat "$OUT/batch${K}_positions.txt"
done
