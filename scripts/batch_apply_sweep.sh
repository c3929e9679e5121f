#!/bin/bash
# Tuning experiment (GPU box): time of one batched factor apply (k = 16 and 8) for launch geometries of fc_nd_block_b.
# usage: scripts/batch_apply_sweep.sh   (writes gpurun_out/r3_apply_sweep.txt)
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3_apply_sweep.txt
: > $OUT
run() {
  echo "== $*" >> $OUT
  env "$@" timeout -k 10 200 python scripts/batch_probe.py --skip-parity --ks 8,16 --steps 100 2>&1 | grep "^k=" >> $OUT
}
run FC_BATCH_CPW=3
run FC_BATCH_CPW=1.5
run FC_BATCH_CPW=6
run FC_BATCH_CPW=12
run FC_BATCH_CG=1
run FC_BATCH_CG=4
cat $OUT
