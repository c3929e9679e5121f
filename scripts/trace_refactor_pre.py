"""What runs between the start of an fc_refactor and its first front kernel (memset of the fronts, permuted matrix copy, scatter): reads the raw kernel trace of scripts/trace_refactor.sh (KEEP_RAW=1)."""
import glob, sys
import pandas as pd
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
t = pd.read_csv(f).sort_values("Start_Timestamp").reset_index(drop=True)
t["name"] = t["Kernel_Name"].str.replace("void ", "").str.slice(0, 40)
i0 = t.index[t["name"].str.startswith("fc_front_scatter")][-1]
s = t.loc[i0 - 6 : i0 + 2].copy()
s["t_us"] = (s["Start_Timestamp"] - t.loc[i0, "Start_Timestamp"]) / 1e3
s["dur_us"] = (s["End_Timestamp"] - s["Start_Timestamp"]) / 1e3
print(s[["name", "t_us", "dur_us"]].to_string())
