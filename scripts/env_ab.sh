#!/bin/bash
# A/B of one environment switch on the four benchmark meshes: VAR=name A=value B=value
out=gpurun_out/env_ab_$VAR.txt
: > $out
for V in $A $B $A $B; do
  export $VAR=$V
  a=$(timeout -k 10 120 python bench.py --no-cpu-baseline --no-large-spmv --no-replicas --steps 3000 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]), round(d["phase_ms_eager"]["sweeps"]*1e3,2))')
  b=$(timeout -k 10 120 python bench.py --refine 1 --no-cpu-baseline --no-large-spmv --no-replicas --steps 1500 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]), round(d["phase_ms_eager"]["sweeps"]*1e3,2))')
  c=$(timeout -k 10 200 python scripts/bench_case.py pinball --steps 2000 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]), round(d["roofline"]["apply_us"],1))')
  d=$(timeout -k 10 200 python scripts/bench_case.py cavity_fine --steps 300 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["roofline"]["apply_us"],1))')
  echo "$VAR=$V: O1 $a | refined O1 $b | pinball $c | cavity_fine $d" >> $out
done
cat $out
