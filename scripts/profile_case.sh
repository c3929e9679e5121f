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
for name, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    c = pd.read_csv(glob.glob(f"{out}/{sub}/**/*_counter_collection.csv", recursive=True)[0])
    c = c[c["Counter_Name"] == name]
    sw = c[c["Kernel_Name"].str.contains("fc_nd_sweep|fc_nd_down_block")]
    d = json.loads(open(f"{out}/bench_{'fetch' if name == 'FETCH_SIZE' else 'write'}.json").read().strip().splitlines()[-1])
    launches = d["roofline"]["launches_per_apply"]
    n_apply = len(sw.groupby("Dispatch_Id")) / launches
    res[name + "_KB_per_apply_raw"] = float(sw["Counter_Value"].sum() / n_apply)
    res["bytes_per_apply_algorithmic"] = d["roofline"]["bytes_per_apply"]
res["bytes_per_apply_counters"] = (2.0 * res["FETCH_SIZE_KB_per_apply_raw"] + res["WRITE_SIZE_KB_per_apply_raw"]) * 1024.0  # gfx950: FETCH_SIZE counts 128-B requests at 64 B
res["bytes_per_apply_counters_uncorrected"] = (res["FETCH_SIZE_KB_per_apply_raw"] + res["WRITE_SIZE_KB_per_apply_raw"]) * 1024.0
json.dump(res, open(out + "/traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf "$OUT/stats" "$OUT/pmc_fetch" "$OUT/pmc_write"
