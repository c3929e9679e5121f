#!/bin/bash
# Per-launch timeline of ONE fc_refactor (the last of scripts/refactor_time.py): kernel, start offset, duration, grid
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/trace_refactor
rm -rf "$OUT" && mkdir -p "$OUT"
rocprofv3 --kernel-trace --output-format csv -d "$OUT/raw" -- python scripts/refactor_time.py ${MESH:-O1} > "$OUT/log.txt" 2> "$OUT/err.txt"
python - <<PY
import glob, pandas as pd
f = glob.glob("$OUT/raw/**/*_kernel_trace.csv", recursive=True)[0]
t = pd.read_csv(f).sort_values("Start_Timestamp").reset_index(drop=True)
t["name"] = t["Kernel_Name"].str.replace("void ", "").str.slice(0, 28)
# the last factorisation: from the last fc_front_scatter to the last fc_fe_export
i0 = t.index[t["name"].str.startswith("fc_front_scatter")][-1]
i1 = t.index[t["name"].str.startswith("fc_fe_export")][-1]
s = t.loc[i0:i1].copy()
s["t_us"] = (s["Start_Timestamp"] - s["Start_Timestamp"].iloc[0]) / 1e3
s["dur_us"] = (s["End_Timestamp"] - s["Start_Timestamp"]) / 1e3
s["gap_us"] = (s["Start_Timestamp"] - s["End_Timestamp"].shift(1)) / 1e3
s["wgs"] = s["Grid_Size_X"] // s["Workgroup_Size_X"] * (s["Grid_Size_Y"] // s["Workgroup_Size_Y"])
s[["name", "t_us", "dur_us", "gap_us", "wgs", "Workgroup_Size_X", "LDS_Block_Size", "VGPR_Count"]].to_csv("$OUT/timeline.csv", index=False, float_format="%.2f")
print(s.groupby("name")["dur_us"].agg(["count", "sum", "mean"]).sort_values("sum", ascending=False).to_string())
print("total span us", s["t_us"].iloc[-1] + s["dur_us"].iloc[-1], "sum of gaps", s["gap_us"].iloc[1:].sum())
PY
[ -n "$KEEP_RAW" ] && python scripts/trace_refactor_pre.py "$OUT/raw"; rm -rf "$OUT/raw"
