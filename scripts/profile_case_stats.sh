#!/bin/bash
# per-kernel times of a large case's closed-loop steps (no counters): scripts/profile_case_stats.sh cavity_fine|pinball
set -e
CASE=${1:-cavity_fine}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profstats_$CASE
rm -rf "$OUT" && mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python scripts/bench_case.py $CASE --steps ${STEPS:-100} --warmup 5 > "$OUT/bench_stats.json" 2> "$OUT/stats.err"
python - <<PY
import glob
import pandas as pd
out = "$OUT"
ks = pd.read_csv(glob.glob(out + "/stats/**/*_kernel_stats.csv", recursive=True)[0])
ks["Name"] = ks["Name"].str.slice(0, 70)
ks.head(24).to_csv(out + "/kernel_stats.csv", index=False)
print(ks[["Name", "Calls", "AverageNs", "Percentage"]].head(24).to_string(index=False))
PY
rm -rf "$OUT/stats"
