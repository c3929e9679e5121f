#!/bin/bash
# rocprofv3 kernel stats of the factorisation-free Krylov mode (100 cylinder steps): scripts/profile_krylov_free.sh -> gpurun_out/r05_kf/
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r05_kf
mkdir -p "$OUT"
python scripts/krylov_free_steps.py 200 gmres > "$OUT/steps_gmres.txt" 2>&1
python scripts/krylov_free_steps.py 200 bicgstab > "$OUT/steps_bicgstab.txt" 2>&1
FC_KRYLOV_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- python scripts/krylov_free_steps.py 100 gmres > "$OUT/steps_under_rocprof.txt" 2>&1
cp $(find "$OUT/prof" -name "*_kernel_stats.csv" | head -1) "$OUT/kernel_stats.csv"
rm -rf "$OUT/prof"
cat "$OUT"/steps_*.txt
head -30 "$OUT/kernel_stats.csv"
