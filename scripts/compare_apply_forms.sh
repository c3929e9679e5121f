#!/bin/bash
set -e
CASE=${1:-cavity_fine}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/forms_$CASE
rm -rf "$OUT" && mkdir -p "$OUT"
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python scripts/compare_apply_forms.py $CASE > "$OUT/run.log" 2> "$OUT/run.err"
python - <<PY
import glob
import numpy as np
import pandas as pd
out = "$OUT"
df = pd.read_csv(glob.glob(out + "/trace/**/*_kernel_trace.csv", recursive=True)[0]).sort_values("Start_Timestamp").reset_index(drop=True)
df["dur"] = df["End_Timestamp"] - df["Start_Timestamp"]
df["k"] = df["Kernel_Name"].str.slice(0, 48)
def positions(mask_first, member):
    sub = df[member].reset_index(drop=True)
    starts = sub.index[mask_first(sub)].values
    segs = [sub.loc[a:b - 1] for a, b in zip(starts[:-1], starts[1:])]
    n = int(np.bincount([len(s) for s in segs]).argmax())
    rows = np.array([s["dur"].values for s in segs if len(s) == n])
    seg = next(s for s in segs if len(s) == n)
    return pd.DataFrame({"kernel": seg["k"].values, "grid": seg["Grid_Size_X"].values, "median_us": np.median(rows, axis=0) / 1e3})
row = df["Kernel_Name"].str.contains("fc_nd_sweep|fc_nd_down_block")
# an apply of the row form starts with the launch that follows a non-sweep kernel
r = df[row | df["Kernel_Name"].str.contains("fc_gather_perm|fc_copy|fc_scatter_perm|fc_spmv")].reset_index(drop=True)
tab = positions(lambda s: s["Kernel_Name"].str.contains("fc_nd_sweep").values & ~np.r_[False, s["Kernel_Name"].str.contains("fc_nd_sweep|fc_nd_down_block").values[:-1]], row | ~row)
tab = tab[tab["kernel"].str.contains("fc_nd_sweep|fc_nd_down_block")]
print(tab.to_string()); print("row form sum", tab["median_us"].sum())
blk = df["Kernel_Name"].str.contains("fc_nd_block_b|fc_nd_fold_b")
sub = df[blk].reset_index(drop=True)
n = len(tab)
# the batched apply: 16-ish launches per apply, the probe runs them back to back
per = None
for cand in range(8, 64):
    if len(sub) % cand == 0 and (sub["k"].values[:cand] == sub["k"].values[cand:2 * cand]).all() and (sub["Grid_Size_X"].values[:cand] == sub["Grid_Size_X"].values[cand:2 * cand]).all():
        per = cand; break
rows = sub["dur"].values.reshape(-1, per)
t2 = pd.DataFrame({"kernel": sub["k"].values[:per], "grid": sub["Grid_Size_X"].values[:per], "median_us": np.median(rows, axis=0) / 1e3})
print(t2.to_string()); print("block form sum", t2["median_us"].sum())
tab.to_csv(out + "/row_form.csv", index=False); t2.to_csv(out + "/block_form.csv", index=False)
PY
rm -rf "$OUT/trace"
