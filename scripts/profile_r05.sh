#!/bin/bash
# Round-5 evidence pass on the GPU box (through gpurun), everything from the final binary:  PART=A|B scripts/profile_r05.sh  -> gpurun_out/r05/*
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r05
PART=${PART:-AB}  # a gpurun call is limited to 20 minutes: PART=A, then PART=B
mkdir -p "$OUT"
(while true; do sleep 60; echo "[$(date +%T)] still profiling"; done) &
HB=$!
trap 'kill $HB 2>/dev/null' EXIT
if [[ $PART == *A* ]]; then
# 1. the default bench line (what the driver records)
python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
echo "bench done"
# 2. single-simulation step on one stream: kernel stats, per-position medians, FETCH / WRITE counter passes
STEPS=600 bash scripts/profile_gpu.sh > "$OUT/profile_gpu.log" 2>&1 || true
cp gpurun_out/profile/kernel_stats.csv "$OUT/kernel_stats.csv" 2>/dev/null || true
cp gpurun_out/profile/sweep_stages.csv "$OUT/step_kernels.csv" 2>/dev/null || true
cp gpurun_out/profile/traffic.json "$OUT/traffic.json" 2>/dev/null || true
cp gpurun_out/profile/bench_stats.json "$OUT/bench_under_rocprof.json" 2>/dev/null || true
echo "step profile done"
# 3. batched step: per-position medians for k = 32 and 16 (one stream, plain launches), timelines of the default step, throughput probe
FC_OVERLAP_TAIL=0 bash scripts/profile_batch32.sh > "$OUT/profile_batch32.log" 2>&1 || true
cp gpurun_out/b32/batch32_positions.csv "$OUT/batch32_positions.csv" 2>/dev/null || true
cp gpurun_out/b32/batch16_positions.csv "$OUT/batch16_positions.csv" 2>/dev/null || true
K=32 bash scripts/profile_batch_timeline.sh > "$OUT/batch32_timeline.txt" 2>&1 || true
K=16 bash scripts/profile_batch_timeline.sh > "$OUT/batch16_timeline.txt" 2>&1 || true
python scripts/batch_probe.py --steps 400 --reps 3 > "$OUT/batch_probe_O1.log" 2>&1 || true
python scripts/host_split_batch.py 16 > "$OUT/host_split_batch16.txt" 2>&1 || true
echo "batch profile done"
# 4. factorisation-free Krylov mode
bash scripts/profile_krylov_free.sh > "$OUT/profile_krylov_free.log" 2>&1 || true
cp gpurun_out/r05_kf/kernel_stats.csv "$OUT/krylov_free_kernel_stats.csv" 2>/dev/null || true
cat gpurun_out/r05_kf/steps_gmres.txt gpurun_out/r05_kf/steps_bicgstab.txt 2>/dev/null | grep -v amdgpu > "$OUT/krylov_free_steps.txt" || true
for m in O1 mesh_middle_gmsh cavity_fine; do python scripts/krylov_free_probe.py $m 2>&1 | grep -v amdgpu >> "$OUT/krylov_free_probe.txt" || true; done
echo "krylov done"
fi
if [[ $PART == *B* ]]; then
# 5. numeric factorisation times, FETCH_SIZE calibration
python scripts/refactor_time.py O1 mesh_middle_gmsh cavity_coarse cavity_fine > "$OUT/refactor_times.txt" 2>&1 || true
bash scripts/fetch_calib.sh > "$OUT/fetch_calib.log" 2>&1 || true
cp gpurun_out/fetch_calib/fetch_calib.csv "$OUT/fetch_calib.csv" 2>/dev/null || true
# 6. long closed-loop runs of configs 5, 3, 4 + kernel stats / sweep traffic / per-launch TB/s on cavity_fine and the pinball
python scripts/bench_case.py pinball --steps 10000 > "$OUT/bench_pinball_10k.json" 2> "$OUT/bench_pinball.err" || true
python scripts/bench_case.py cavity_fine --steps 1000 > "$OUT/bench_cavity_fine.json" 2> "$OUT/bench_cavity_fine.err" || true
python scripts/bench_case.py refined1 --steps 2000 > "$OUT/bench_refined1.json" 2> "$OUT/bench_refined1.err" || true
for c in cavity_fine pinball; do
STEPS=60 bash scripts/profile_case.sh $c > "$OUT/profile_$c.log" 2>&1 || true
cp gpurun_out/prof_$c/kernel_stats.csv "$OUT/${c}_kernel_stats.csv" 2>/dev/null || true
cp gpurun_out/prof_$c/traffic.json "$OUT/${c}_sweep_traffic.json" 2>/dev/null || true
cp gpurun_out/prof_$c/launch_positions.csv "$OUT/${c}_launch_positions.csv" 2>/dev/null || true
done
echo "cases done"
# 7. rehearsal of the N > 1 bench path on this one GPU: 8 thread ranks
FC_BENCH_THREAD_RANKS=8 python bench.py --gpus 8 --steps 50 --warmup 5 > "$OUT/bench_threads8_rehearsal.json" 2> "$OUT/bench_threads8.err" || true
fi
echo "all done"
ls -la "$OUT"
