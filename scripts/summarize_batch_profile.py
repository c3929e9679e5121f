"""Median duration of every launch position inside one batched step (segments between consecutive fc_rhs_elem_b
launches of a rocprofv3 --kernel-trace run).  usage: summarize_batch_profile.py <rocprof output dir> [out.csv]"""
import glob
import sys

import numpy as np
import pandas as pd

d = sys.argv[1]
tr = glob.glob(f"{d}/**/*_kernel_trace.csv", recursive=True)[0]
df = pd.read_csv(tr).sort_values("Start_Timestamp").reset_index(drop=True)
df["dur"] = df["End_Timestamp"] - df["Start_Timestamp"]
idx = df.index[df["Kernel_Name"].str.contains("fc_rhs_elem_b")].values
segs = [df.loc[a : b - 1] for a, b in zip(idx[20:-1], idx[21:])]
lens = np.array([len(s) for s in segs])
n = int(np.bincount(lens).argmax())
rows = np.array([s["dur"].values for s in segs if len(s) == n])
gaps = np.array([(s["Start_Timestamp"].values[1:] - s["End_Timestamp"].values[:-1]) for s in segs if len(s) == n])
seg = next(s for s in segs if len(s) == n)
tab = pd.DataFrame({"position": range(n), "kernel": [k[:48] for k in seg["Kernel_Name"]], "grid": seg["Grid_Size_X"].values // 256,
                    "vgpr": seg["VGPR_Count"].values, "median_ns": np.median(rows, axis=0), "gap_before_ns": np.r_[0, np.median(gaps, axis=0)]})
print(tab.to_string(index=False))
print("kernel sum per batched step [us]:", tab["median_ns"].sum() / 1e3, " gaps [us]:", tab["gap_before_ns"].sum() / 1e3)
print("step period [us]:", np.median(np.diff(df.loc[idx[20:], "Start_Timestamp"].values)) / 1e3)
if len(sys.argv) > 2:
    tab.to_csv(sys.argv[2], index=False)
