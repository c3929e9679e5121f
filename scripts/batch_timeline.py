"""Timeline of the DEFAULT batched step (overlapped tail, graph replay) from a rocprofv3 --kernel-trace run:
per step (fc_early_b to fc_early_b) the period, the busy time of the main chain, the gaps between its kernels and what ran beside it.
usage: batch_timeline.py <rocprof output dir>"""
import glob
import sys

import numpy as np
import pandas as pd

tr = glob.glob(f"{sys.argv[1]}/**/*_kernel_trace.csv", recursive=True)[0]
df = pd.read_csv(tr).sort_values("Start_Timestamp").reset_index(drop=True)
df["name"] = df["Kernel_Name"].str.replace(r"^void ", "", regex=True).str.slice(0, 22)
side = df["name"].str.contains("fc_wait_solved_b|fc_tail_b|fc_final_late_b")
main = df[~side].reset_index(drop=True)
idx = main.index[main["name"].str.contains("fc_early_b")].values
rows = []
for a, b in zip(idx[30:-2], idx[31:-1]):
    seg = main.loc[a + 1 : b]
    busy = (seg["End_Timestamp"] - seg["Start_Timestamp"]).sum()
    period = main.loc[b, "End_Timestamp"] - main.loc[a, "End_Timestamp"]
    first_gap = seg["Start_Timestamp"].iloc[0] - main.loc[a, "End_Timestamp"]
    gaps = (seg["Start_Timestamp"].values[1:] - seg["End_Timestamp"].values[:-1]).sum()
    rows.append((period, busy, first_gap, gaps, len(seg)))
r = np.array(rows, dtype=float)
print(f"steps {len(r)}: period {np.median(r[:,0])/1e3:.1f} us, main-chain busy {np.median(r[:,1])/1e3:.1f} us, gap after fc_early_b (host turn-around) "
      f"{np.median(r[:,2])/1e3:.1f} us, other gaps {np.median(r[:,3])/1e3:.1f} us, kernels {int(np.median(r[:,4]))}")
a, b = idx[40], idx[41]
t0 = main.loc[a, "End_Timestamp"]
seg = df[(df["Start_Timestamp"] >= main.loc[a, "Start_Timestamp"]) & (df["End_Timestamp"] <= main.loc[b, "End_Timestamp"])]
for _, k in seg.iterrows():
    print(f"  {(k['Start_Timestamp'] - t0) / 1e3:8.1f} .. {(k['End_Timestamp'] - t0) / 1e3:8.1f}  {(k['End_Timestamp'] - k['Start_Timestamp']) / 1e3:6.1f} us  {k['name']}")
