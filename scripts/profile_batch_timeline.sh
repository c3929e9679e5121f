#!/bin/bash
# timeline of the default (overlapped, graph-replayed) batched step: K=32 scripts/profile_batch_timeline.sh
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
K=${K:-32}
OUT=gpurun_out/btl; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d "$OUT/prof" -- python scripts/batch_probe.py --skip-parity --ks $K --steps 150 $PROBE_ARGS > "$OUT/probe.log" 2>&1
python scripts/batch_timeline.py "$OUT/prof" > "$OUT/timeline_k$K.txt"
rm -rf "$OUT/prof"
cat "$OUT/timeline_k$K.txt"
