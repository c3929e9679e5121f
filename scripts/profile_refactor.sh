#!/bin/bash
# Kernel-level breakdown of the device factorisation (fc_refactor): scripts/profile_refactor.sh
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_refactor
rm -rf "$OUT" && mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python scripts/refactor_time.py ${MESH:-O1} > "$OUT/log.txt" 2> "$OUT/err.txt"
python - <<PY
import glob, pandas as pd
f = glob.glob("$OUT/stats/**/*_kernel_stats.csv", recursive=True)[0]
ks = pd.read_csv(f)
ks["Name"] = ks["Name"].str.slice(0, 90)
print(ks[["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"]].head(25).to_string(index=False))
ks.head(40).to_csv("$OUT/kernel_stats.csv", index=False)
PY
rm -rf "$OUT/stats"
tail -5 "$OUT/log.txt"
