#!/bin/bash
# Per-launch-position hardware counters of one step on the refined mesh (separate --pmc passes, kernel trace only).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_refined
rm -rf "$OUT" && mkdir -p "$OUT"
ARGS="--refine ${REFINE:-1} --steps 40 --warmup 10 --no-cpu-baseline --no-large-spmv"
i=0
for set in "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "TCC_EA0_RDREQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/p$i" -- python bench.py $ARGS > "$OUT/bench$i.json" 2> "$OUT/err$i.txt"
  echo "pass $i done"
done
python - <<PY
import glob
import numpy as np
import pandas as pd
rows = {}
for f in glob.glob("$OUT/p*/**/*_counter_collection.csv", recursive=True):
    c = pd.read_csv(f)
    c = c.sort_values("Dispatch_Id")
    disp = c.drop_duplicates("Dispatch_Id")[["Dispatch_Id", "Kernel_Name"]].reset_index(drop=True)
    idx = disp.index[disp["Kernel_Name"].str.contains("fc_rhs_elem")].values
    segs = [(a, b) for a, b in zip(idx[10:-1], idx[11:])]
    n = int(np.bincount([b - a for a, b in segs]).argmax())
    segs = [(a, b) for a, b in segs if b - a == n]
    for name, grp in c.groupby("Counter_Name"):
        v = grp.groupby("Dispatch_Id")["Counter_Value"].sum().reindex(disp["Dispatch_Id"]).values
        rows[name] = np.median(np.array([v[a:b] for a, b in segs]), axis=0)
    kern = [k[:34] for k in disp["Kernel_Name"].values[segs[0][0]:segs[0][1]]]
tab = pd.DataFrame(rows)
tab.insert(0, "kernel", kern)
pd.set_option("display.width", 250)
print(tab.to_string())
tab.to_csv("$OUT/pmc_by_position.csv", index=False)
PY
rm -rf "$OUT"/p1 "$OUT"/p2 "$OUT"/p3
