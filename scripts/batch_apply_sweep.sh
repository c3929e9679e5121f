#!/bin/bash
# Tuning experiment (GPU box): time of one batched factor apply for launch geometries of fc_nd_block_b.
# usage: scripts/batch_apply_sweep.sh [probe args]   (writes gpurun_out/r3_apply_sweep.txt)
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3_apply_sweep.txt
: > $OUT
run() {
  echo "== $*" >> $OUT
  env "$@" timeout -k 10 300 python scripts/batch_probe.py --skip-parity --ks 16 --steps 60 $ARGS 2>&1 | grep "^k=" >> $OUT
}
ARGS="$*"
run FC_BATCH_CPW=1.5
run FC_BATCH_CPW=3
run FC_BATCH_CPW=6
run FC_BATCH_CPW=12
run FC_BATCH_CPW=24
run FC_BATCH_CG=1
cat $OUT
