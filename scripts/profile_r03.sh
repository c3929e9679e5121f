#!/bin/bash
# Round-3 evidence pass on the GPU box (through gpurun): everything DESIGN.md / profiles/README.md quote, from the final binary.
#   scripts/profile_r03.sh            -> gpurun_out/r03/*
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03
rm -rf "$OUT" && mkdir -p "$OUT"
(while true; do sleep 60; echo "[$(date +%T)] still profiling"; done) &
HB=$!
trap 'kill $HB 2>/dev/null' EXIT
# 1. the default bench line (what the driver records), incl. the batched-replicas section
python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
echo "bench done"
# 2. single-simulation step: kernel stats, per-position medians, FETCH / WRITE counter passes (scripts/profile_gpu.sh)
STEPS=600 bash scripts/profile_gpu.sh > "$OUT/profile_gpu.log" 2>&1 || true
cp gpurun_out/profile/kernel_stats.csv "$OUT/kernel_stats.csv" 2>/dev/null || true
cp gpurun_out/profile/sweep_stages.csv "$OUT/step_kernels.csv" 2>/dev/null || true
cp gpurun_out/profile/traffic.json "$OUT/traffic.json" 2>/dev/null || true
cp gpurun_out/profile/bench_stats.json "$OUT/bench_under_rocprof.json" 2>/dev/null || true
echo "step profile done"
# 3. batched step: per-position medians for k = 16 and k = 8 (plain launches so that every kernel is a trace record)
for K in 16 8; do
  FC_BATCH_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_b$K" -- python scripts/batch_probe.py --skip-parity --ks $K --steps 200 > "$OUT/batch_probe_rocprof_k$K.log" 2>&1
  python scripts/summarize_batch_profile.py "$OUT/prof_b$K" "$OUT/batch${K}_positions.csv" > "$OUT/batch${K}_positions.txt"
  cp $(find "$OUT/prof_b$K" -name "*_kernel_stats.csv" | head -1) "$OUT/batch${K}_kernel_stats.csv"
  rm -rf "$OUT/prof_b$K"
done
python scripts/batch_probe.py --steps 400 > "$OUT/batch_probe_O1.log" 2>&1
python scripts/batch_probe.py --skip-parity --refine 1 --steps 100 > "$OUT/batch_probe_refined1.log" 2>&1
echo "batch profile done"
# 4. numeric factorisation times (every figure DESIGN quotes) + kernel breakdown on O1
python scripts/refactor_time.py O1 mesh_middle_gmsh cavity_coarse cavity_fine > "$OUT/refactor_times.txt" 2>&1
bash scripts/profile_refactor.sh > "$OUT/refactor_kernel_stats.txt" 2>&1 || true
cp gpurun_out/prof_refactor/kernel_stats.csv "$OUT/refactor_kernel_stats.csv" 2>/dev/null || true
# per-launch timeline of one factorisation (which launches form the dependent chain): O1 and cavity_fine
for M in O1 cavity_fine; do
  MESH=$M bash scripts/trace_refactor.sh > "$OUT/refactor_timeline_$M.txt" 2>&1 || true
  cp gpurun_out/trace_refactor/timeline.csv "$OUT/refactor_timeline_$M.csv" 2>/dev/null || true
done
echo "refactor done"
# 5. closed-loop throughput runs of configs 5 and 3 + the sweeps' traffic on cavity_fine
python scripts/bench_case.py pinball --steps 10000 > "$OUT/bench_pinball_10k.json" 2> "$OUT/bench_pinball.err"
python scripts/bench_case.py cavity_fine --steps 1000 > "$OUT/bench_cavity_fine.json" 2> "$OUT/bench_cavity_fine.err"
STEPS=60 bash scripts/profile_case.sh cavity_fine > "$OUT/profile_cavity_fine.log" 2>&1 || true
cp gpurun_out/prof_cavity_fine/kernel_stats.csv "$OUT/cavity_fine_kernel_stats.csv" 2>/dev/null || true
cp gpurun_out/prof_cavity_fine/traffic.json "$OUT/cavity_fine_sweep_traffic.json" 2>/dev/null || true
echo "cases done"
# 6. SpMV traffic on the cavity_fine-size matrix: 64-B read requests vs 32-B read requests in ONE counter pass
timeout -k 10 600 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d "$OUT/pmc_spmv" -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-replicas > "$OUT/bench_spmv_pmc.json" 2> "$OUT/spmv_pmc.err" || true
python - <<PY || true
import glob, json, pandas as pd
out = "$OUT"
f = glob.glob(out + "/pmc_spmv/**/*_counter_collection.csv", recursive=True)
if f:
    c = pd.read_csv(f[0])
    c = c[c["Kernel_Name"].str.contains("fc_spmv_csr<8, 0>")]
    res = {}
    for grid, grp in c.groupby("Grid_Size"):
        piv = grp.pivot_table(index="Dispatch_Id", columns="Counter_Name", values="Counter_Value", aggfunc="sum")
        m = piv.mean()
        rd, rd32 = float(m.get("TCC_EA0_RDREQ_sum", float("nan"))), float(m.get("TCC_EA0_RDREQ_32B_sum", float("nan")))
        res[str(int(grid))] = {"launches": int(len(piv)), "TCC_EA0_RDREQ": rd, "TCC_EA0_RDREQ_32B": rd32,
                               "bytes_if_requests_are_64B_or_32B": 64.0 * (rd - rd32) + 32.0 * rd32, "bytes_if_non_32B_requests_are_128B": 128.0 * (rd - rd32) + 32.0 * rd32}
    d = json.loads(open(out + "/bench_spmv_pmc.json").read().strip().splitlines()[-1])
    res["algorithmic_bytes"] = {k: v["bytes"] for k, v in d["spmv"].items()}
    json.dump(res, open(out + "/spmv_traffic.json", "w"), indent=1)
    print(json.dumps(res, indent=1))
PY
rm -rf "$OUT/pmc_spmv"
echo "all done"
ls -la "$OUT"
