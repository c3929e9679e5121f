#!/bin/bash
# FETCH_SIZE calibration for 8-B / 16-B per lane and 8- / 16-lane row access (scripts/micro/fetch_calib.hip): scripts/fetch_calib.sh -> gpurun_out/fetch_calib/
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/fetch_calib; rm -rf $OUT; mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 scripts/micro/fetch_calib.hip -o /tmp/fetch_calib
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/prof" -- /tmp/fetch_calib > "$OUT/run.txt" 2>&1
python - "$OUT" <<'PY'
import glob, sys
import pandas as pd
out = sys.argv[1]
f = glob.glob(f"{out}/prof/**/*counter_collection.csv", recursive=True)[0]
d = pd.read_csv(f)
nbytes = 2**31
rows = []
for name, g in d[d["Counter_Name"] == "FETCH_SIZE"].groupby("Kernel_Name"):
    v = float(g["Counter_Value"].sum())
    rows.append((name.split("(")[0], v, v * 1024, nbytes / (v * 1024)))
tab = pd.DataFrame(rows, columns=["kernel", "FETCH_SIZE_raw_KB", "FETCH_SIZE_bytes", "true_bytes_over_counter"])
print(tab.to_string(index=False))
tab.to_csv(f"{out}/fetch_calib.csv", index=False)
PY
rm -rf "$OUT/prof"
