#!/bin/bash
# Kernel stats + HBM traffic counters of the factor sweeps on a large case: scripts/profile_case.sh cavity_fine|pinball
set -e
CASE=${1:-cavity_fine}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$CASE
rm -rf "$OUT" && mkdir -p "$OUT"
(while true; do sleep 60; echo "[$(date +%T)] still profiling"; done) &
HB=$!
trap 'kill $HB 2>/dev/null' EXIT
ARGS="$CASE --steps ${STEPS:-60} --warmup 5"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python scripts/bench_case.py $ARGS > "$OUT/bench_stats.json" 2> "$OUT/stats.err"
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python scripts/bench_case.py $CASE --steps 20 --warmup 3 > "$OUT/bench_fetch.json" 2> "$OUT/fetch.err"
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python scripts/bench_case.py $CASE --steps 20 --warmup 3 > "$OUT/bench_write.json" 2> "$OUT/write.err"
python - <<PY
import glob, json
import pandas as pd
out = "$OUT"
ks = pd.read_csv(glob.glob(out + "/stats/**/*_kernel_stats.csv", recursive=True)[0])
ks["Name"] = ks["Name"].str.slice(0, 70)
ks.head(16).to_csv(out + "/kernel_stats.csv", index=False)
print(ks[["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"]].head(16).to_string(index=False))
res = {}
per_kernel = {}
for name, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    c = pd.read_csv(glob.glob(f"{out}/{sub}/**/*_counter_collection.csv", recursive=True)[0])
    c = c[c["Counter_Name"] == name]
    # every launch of a factor apply (segment sweeps, LDS-tiled blocks, the column form's fold launches); applies of the run (time steps,
    # base-flow iterations, acceptance solves alike) = those launches / launches per apply as the library reports them
    sw = c[c["Kernel_Name"].str.contains("fc_nd_sweep|fc_nd_down_block|fc_nd_flat_block|fc_nd_fold1")]
    d = json.loads(open(f"{out}/bench_{'fetch' if name == 'FETCH_SIZE' else 'write'}.json").read().strip().splitlines()[-1])
    res["launches_per_apply"] = d["roofline"]["launches_per_apply"]
    n_apply = len(sw.groupby("Dispatch_Id")) / res["launches_per_apply"]
    res[name + "_KB_per_apply_raw"] = float(sw["Counter_Value"].sum() / n_apply)
    res["bytes_per_apply_algorithmic"] = d["roofline"]["bytes_per_apply"]
    for kname, g in sw.groupby(sw["Kernel_Name"].str.replace(r"^void ", "", regex=True).str.slice(0, 40)):
        per_kernel.setdefault(kname, {})[name + "_KB_per_apply_raw"] = float(g["Counter_Value"].sum() / n_apply)
# gfx950: FETCH_SIZE tallies the 128-B requests at 64 B -- a factor 2.00 for every access width these kernels use (8 B and 16 B per lane,
# 8- and 16-lane rows, nontemporal or not: scripts/fetch_calib.sh, profiles/r05_fetch_calib.csv); WRITE_SIZE reads true
# per launch position of an apply (kernel, grid): median duration and median bytes read (FETCH_SIZE x 2) -> TB/s of every sweep launch
try:
    cf = pd.read_csv(glob.glob(f"{out}/pmc_fetch/**/*_counter_collection.csv", recursive=True)[0])
    cf = cf[cf["Counter_Name"] == "FETCH_SIZE"]
    tf = pd.read_csv(glob.glob(f"{out}/pmc_fetch/**/*_kernel_trace.csv", recursive=True)[0])
    tf["dur"] = tf["End_Timestamp"] - tf["Start_Timestamp"]
    j = cf.groupby("Dispatch_Id")["Counter_Value"].sum().rename("fetch_kb").to_frame().join(tf.set_index("Dispatch_Id")[["Kernel_Name", "Grid_Size_X", "dur"]], how="inner")
    j = j[j["Kernel_Name"].str.contains("fc_nd_sweep|fc_nd_down_block|fc_nd_flat_block|fc_nd_fold1|fc_tail|fc_rhs")]
    j["kernel"] = j["Kernel_Name"].str.replace(r"^void ", "", regex=True).str.slice(0, 34)
    pos = j.groupby(["kernel", "Grid_Size_X"]).agg(n=("dur", "size"), us=("dur", lambda v: v.median() / 1e3), MB=("fetch_kb", lambda v: 2.0 * v.median() * 1024 / 1e6)).reset_index()
    pos["TB/s"] = pos["MB"] / pos["us"]
    pos = pos[pos["n"] >= 5].sort_values("us", ascending=False)
    pos.to_csv(out + "/launch_positions.csv", index=False)
    print(pos.to_string(index=False))
    print("sum of medians [us]:", pos[pos["kernel"].str.contains("fc_nd")]["us"].sum(), " MB:", pos[pos["kernel"].str.contains("fc_nd")]["MB"].sum())
except Exception as err:
    print("launch positions failed:", err)
res["bytes_per_apply_counters"] = (2.0 * res["FETCH_SIZE_KB_per_apply_raw"] + res["WRITE_SIZE_KB_per_apply_raw"]) * 1024.0
res["counters_over_algorithmic"] = res["bytes_per_apply_counters"] / res["bytes_per_apply_algorithmic"]
res["per_kernel"] = per_kernel
json.dump(res, open(out + "/traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf "$OUT/stats" "$OUT/pmc_fetch" "$OUT/pmc_write"
