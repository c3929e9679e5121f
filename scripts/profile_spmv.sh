#!/bin/bash
# HBM traffic of the stand-alone SpMV launches of bench.py (O1 matrix, cache resident, and the cavity_fine-sized
# matrix, HBM streaming): FETCH_SIZE pass -> gpurun_out/profile_spmv/spmv_traffic.json
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profile_spmv
rm -rf "$OUT" && mkdir -p "$OUT"
# (FC_HOST_FACTOR was removed in round 3: the factorisation always runs on the device)
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/err.txt"
python - <<PY
import glob, json, pandas as pd
f = glob.glob("$OUT/pmc_fetch/**/*_counter_collection.csv", recursive=True)[0]
c = pd.read_csv(f)
c = c[(c["Counter_Name"] == "FETCH_SIZE") & c["Kernel_Name"].str.contains("fc_spmv_csr<8, 0>")]
# two populations of launches: the O1 matrix and the large synthetic one (by grid size)
out = {}
for grid, grp in c.groupby("Grid_Size"):
    out[str(int(grid))] = {"launches": int(len(grp)), "FETCH_KB_raw_mean": float(grp["Counter_Value"].mean()),
                           "bytes_doubled": float(grp["Counter_Value"].mean()) * 2 * 1024}
d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
out["algorithmic_bytes"] = {k: v["bytes"] for k, v in d["spmv"].items()}
json.dump(out, open("$OUT/spmv_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf "$OUT/pmc_fetch"
