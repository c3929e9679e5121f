#!/bin/bash
# steps/s (and sweep / apply time) of the four benchmark meshes with the current binary; REPS passes
out=gpurun_out/four_meshes.txt
: > $out
for i in $(seq 1 ${REPS:-2}); do
  a=$(timeout -k 10 120 python bench.py --no-cpu-baseline --no-large-spmv --no-replicas --steps 3000 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]), round(d["phase_ms_eager"]["sweeps"]*1e3,2))')
  b=$(timeout -k 10 120 python bench.py --refine 1 --no-cpu-baseline --no-large-spmv --no-replicas --steps 1500 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]), round(d["phase_ms_eager"]["sweeps"]*1e3,2))')
  c=$(timeout -k 10 200 python scripts/bench_case.py pinball --steps 2000 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]), round(d["roofline"]["apply_us"],1))')
  d=$(timeout -k 10 200 python scripts/bench_case.py cavity_fine --steps 300 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["roofline"]["apply_us"],1))')
  echo "${TAG:-run} $i: O1 $a | refined O1 $b | pinball $c | cavity_fine $d" >> $out
done
cat $out
