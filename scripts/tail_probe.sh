#!/bin/bash
# fc_tail / fc_final / fc_spmv durations of one case from a kernel trace: scripts/tail_probe.sh cavity_fine [steps]
# (the last third of the fc_tail launches runs without the energy cells: scripts/tail_probe.py)
set -e
CASE=${1:-cavity_fine}
STEPS=${2:-60}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/tailprobe_${CASE}${TAG:+_$TAG}
rm -rf "$OUT" && mkdir -p "$OUT"
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python scripts/tail_probe.py "$CASE" --steps "$STEPS" > "$OUT/probe.log" 2> "$OUT/probe.err"
python - <<PY
import glob
import pandas as pd
out, steps = "$OUT", $STEPS
t = pd.read_csv(glob.glob(out + "/trace/**/*_kernel_trace.csv", recursive=True)[0]).sort_values("Start_Timestamp")
t["us"] = (t["End_Timestamp"] - t["Start_Timestamp"]) / 1e3
t["k"] = t["Kernel_Name"].str.extract(r"((?:void )?fc_[a-z_0-9]+)")[0].str.replace("void ", "")
tail = t[t["k"] == "fc_tail"]["us"].to_numpy()
lines = [open(out + "/probe.log").read().strip()]
n = len(tail)
lines.append(f"fc_tail launches {n}: with energy cells median {pd.Series(tail[n - 2 * steps : n - steps]).median():.1f} us, without {pd.Series(tail[n - steps :]).median():.1f} us")
t["kk"] = t["Kernel_Name"].str.extract(r"((?:void )?fc_[a-z_0-9]+(?:<[^>]*>)?)")[0].str.replace("void ", "")
last = t.iloc[-(len(t) // 3):]  # the stepping part of the run (setup kernels come first)
for k, u in last.groupby("kk")["us"]:
    if len(u) >= 10:
        lines.append(f"{k}: {len(u)} launches, median {u.median():.1f} us, mean {u.mean():.1f} us")
open(out + "/summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
rm -rf "$OUT/trace"
