#!/bin/bash
# Round-4 evidence pass on the GPU box (through gpurun), everything from the final binary:  FC_COMMIT=<hash> scripts/profile_r04.sh  -> gpurun_out/r04/*
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04
PART=${PART:-AB}  # a gpurun call is limited to 20 minutes: PART=A (steps 1-4), then PART=B (steps 5-7)
mkdir -p "$OUT"
(while true; do sleep 60; echo "[$(date +%T)] still profiling"; done) &
HB=$!
trap 'kill $HB 2>/dev/null' EXIT
if [[ $PART == *A* ]]; then
# 1. the default bench line (what the driver records): headline, roofline, cpu_baseline, replicas, spmv, other_configs (configs 4 / 5 / 3)
python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
echo "bench done"
# 2. single-simulation step: kernel stats, per-position medians, FETCH / WRITE counter passes (one-stream step: scripts/profile_gpu.sh)
STEPS=600 bash scripts/profile_gpu.sh > "$OUT/profile_gpu.log" 2>&1 || true
cp gpurun_out/profile/kernel_stats.csv "$OUT/kernel_stats.csv" 2>/dev/null || true
cp gpurun_out/profile/sweep_stages.csv "$OUT/step_kernels.csv" 2>/dev/null || true
cp gpurun_out/profile/traffic.json "$OUT/traffic.json" 2>/dev/null || true
cp gpurun_out/profile/bench_stats.json "$OUT/bench_under_rocprof.json" 2>/dev/null || true
echo "step profile done"
# 3. the overlapped step as it runs by default: kernel stats of both streams
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_overlap" -- python bench.py --steps 600 --warmup 20 --no-cpu-baseline --no-large-spmv --no-replicas --no-other-configs > "$OUT/bench_overlap_under_rocprof.json" 2> "$OUT/overlap.err" || true
cp $(find "$OUT/prof_overlap" -name "*_kernel_stats.csv" | head -1) "$OUT/overlap_kernel_stats.csv" 2>/dev/null || true
rm -rf "$OUT/prof_overlap"
# 4. batched step: per-position medians for k = 16 and k = 8 (one-stream, plain launches) and the throughput probe (default: overlapped, graphs)
FC_OVERLAP_TAIL=0 bash scripts/profile_batch.sh "$OUT/batch" > "$OUT/profile_batch.log" 2>&1 || true
python scripts/batch_probe.py --skip-parity --steps 400 > "$OUT/batch_probe_O1_overlapped.log" 2>&1 || true
echo "batch profile done"
fi
if [[ $PART == *B* ]]; then
# 5. numeric factorisation times
python scripts/refactor_time.py O1 mesh_middle_gmsh cavity_coarse cavity_fine > "$OUT/refactor_times.txt" 2>&1 || true
# 6. long closed-loop runs of configs 5 and 3 + kernel stats / sweep traffic on cavity_fine
python scripts/bench_case.py pinball --steps 10000 > "$OUT/bench_pinball_10k.json" 2> "$OUT/bench_pinball.err" || true
python scripts/bench_case.py cavity_fine --steps 1000 > "$OUT/bench_cavity_fine.json" 2> "$OUT/bench_cavity_fine.err" || true
STEPS=60 bash scripts/profile_case.sh cavity_fine > "$OUT/profile_cavity_fine.log" 2>&1 || true
cp gpurun_out/prof_cavity_fine/kernel_stats.csv "$OUT/cavity_fine_kernel_stats.csv" 2>/dev/null || true
cp gpurun_out/prof_cavity_fine/traffic.json "$OUT/cavity_fine_sweep_traffic.json" 2>/dev/null || true
echo "cases done"
# 7. rehearsals of the N > 1 bench path on this one GPU: 8 thread ranks; 4 process ranks over gloo
FC_BENCH_THREAD_RANKS=8 python bench.py --gpus 8 --steps 50 --warmup 5 > "$OUT/bench_threads8_rehearsal.json" 2> "$OUT/bench_threads8.err" || true
FC_BENCH_SAME_DEVICE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 4 --steps 50 --warmup 5 --no-extras > "$OUT/bench_gloo4_rehearsal.json" 2> "$OUT/bench_gloo4.err" || true
fi
echo "all done"
ls -la "$OUT"
