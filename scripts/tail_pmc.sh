#!/bin/bash
# HBM traffic of fc_tail (and of the un-fused residual SpMV) on one case: scripts/tail_pmc.sh cavity_fine
set -e
CASE=${1:-cavity_fine}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/tailpmc_${CASE}${TAG:+_$TAG}
rm -rf "$OUT" && mkdir -p "$OUT"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_$C" -- python scripts/tail_probe.py "$CASE" --steps 8 > "$OUT/probe_$C.log" 2> "$OUT/probe_$C.err"
done
python - <<PY
import glob
import pandas as pd
out = "$OUT"
rows = []
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    c = pd.read_csv(glob.glob(f"{out}/pmc_{name}/**/*_counter_collection.csv", recursive=True)[0])
    c = c[c["Counter_Name"] == name]
    c["k"] = c["Kernel_Name"].str.extract(r"((?:void )?fc_[a-z_0-9]+(?:<[^>]*>)?)")[0].str.replace("void ", "")
    g = c.groupby(["k", "Dispatch_Id"])["Counter_Value"].sum().groupby("k").median()
    rows.append(g.rename(name + "_KB_per_launch"))
t = pd.concat(rows, axis=1)
t["bytes_gfx950"] = (2.0 * t["FETCH_SIZE_KB_per_launch"] + t["WRITE_SIZE_KB_per_launch"]) * 1024.0  # FETCH_SIZE counts 128-B requests at 64 B
t["bytes_uncorrected"] = (t["FETCH_SIZE_KB_per_launch"] + t["WRITE_SIZE_KB_per_launch"]) * 1024.0
t.to_csv(out + "/traffic_by_kernel.csv")
print(t.to_string())
PY
rm -rf "$OUT"/pmc_*
