"""Condense rocprofv3 output (scripts/profile_gpu.sh) into small files for profiles/.

  <out>/kernel_stats.csv        per-kernel calls / total / average ns (from --stats)
  <out>/sweep_stages.csv        median duration of every launch position inside one step
  <out>/traffic.json            FETCH_SIZE / WRITE_SIZE per launch for the hot kernels (raw counter
                                values in KB as rocprofv3 reports them, and bytes)
"""
import glob
import json
import sys
from pathlib import Path

import numpy as np
import pandas as pd

out = Path(sys.argv[1])


def one(pattern):
    m = glob.glob(str(out / pattern), recursive=True)
    return m[0] if m else None


st = one("stats/**/*_kernel_stats.csv")
if st:
    ks = pd.read_csv(st)
    ks.to_csv(out / "kernel_stats.csv", index=False)
    print(ks[["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"]].to_string(index=False, max_colwidth=60))
tr = one("stats/**/*_kernel_trace.csv")
if tr:
    df = pd.read_csv(tr)
    df["dur"] = df["End_Timestamp"] - df["Start_Timestamp"]
    idx = df.index[df["Kernel_Name"].str.contains("fc_rhs_elem")].values
    segs = [df.loc[a : b - 1] for a, b in zip(idx[50:-1], idx[51:])]
    lens = np.array([len(sg) for sg in segs])
    n = int(np.bincount(lens).argmax()) if len(lens) else 0
    rows = [sg["dur"].values for sg in segs if len(sg) == n]
    if rows:
        med = np.median(np.array(rows), axis=0)
        seg = next(sg for sg in segs if len(sg) == n)
        tab = pd.DataFrame({"position": range(n), "kernel": [k[:60] for k in seg["Kernel_Name"]], "grid": seg["Grid_Size_X"].values,
                            "vgpr": seg["VGPR_Count"].values, "median_ns": med})
        tab.to_csv(out / "sweep_stages.csv", index=False)
        print(tab.to_string(index=False))
        print("sum of kernel medians per step [us]:", med.sum() / 1e3)
        print("step period (profiled) [us]:", np.median(np.diff(df.loc[idx[50:], "Start_Timestamp"].values)) / 1e3)

traffic = {}
for name, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    f = one(f"{sub}/**/*_counter_collection.csv")
    if not f:
        continue
    c = pd.read_csv(f)
    c = c[c["Counter_Name"] == name]
    for kern, grp in c.groupby(c["Kernel_Name"].str.slice(0, 40)):
        d = traffic.setdefault(kern, {})
        d[name + "_KB_per_launch"] = float(grp["Counter_Value"].mean())
        d[name + "_launches"] = int(len(grp))
if traffic:
    # the factor sweeps of one apply: segment kernel (up levels, root) + tiled block kernel (down levels)
    sweeps = {k: v for k, v in traffic.items() if "fc_nd_sweep" in k or "fc_nd_down_block" in k or "fc_nd_flat_block" in k}
    tot_f = sum(v.get("FETCH_SIZE_KB_per_launch", 0) * v.get("FETCH_SIZE_launches", 0) for v in sweeps.values())
    n_f = sum(v.get("FETCH_SIZE_launches", 0) for v in sweeps.values())
    tot_w = sum(v.get("WRITE_SIZE_KB_per_launch", 0) * v.get("WRITE_SIZE_launches", 0) for v in sweeps.values())
    n_w = sum(v.get("WRITE_SIZE_launches", 0) for v in sweeps.values())
    summary = {
        "per_kernel": traffic,
        "fc_nd_sweep_FETCH_KB_per_launch_raw": tot_f / max(n_f, 1),
        "fc_nd_sweep_WRITE_KB_per_launch_raw": tot_w / max(n_w, 1),
        # MI355X_MICROARCH.md §HBM: FETCH_SIZE = TCC_EA0_RDREQ x 64 B counts 128-B requests at 64 B for
        # wide coalesced streams -> doubled; WRITE_SIZE is exact.  Our loads are 8 B/lane (uncalibrated
        # width): both raw and doubled figures are kept.
        "fc_nd_sweep_bytes_per_launch": (2.0 * tot_f / max(n_f, 1) + tot_w / max(n_w, 1)) * 1024.0,
        "fc_nd_sweep_bytes_per_launch_uncorrected": (tot_f / max(n_f, 1) + tot_w / max(n_w, 1)) * 1024.0,
    }
    (out / "traffic.json").write_text(json.dumps(summary, indent=1))
    print(json.dumps({k: v for k, v in summary.items() if k != "per_kernel"}, indent=1))
