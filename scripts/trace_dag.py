"""Timeline of one traced one-launch factor apply (fc_debug_trace_apply): per tree-level stage, when its
workgroups started, how long they waited for their dependencies, how long the products and the store drain took.
    python scripts/trace_dag.py [out.npz]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402
from flowcontrol_amd._lib import SLOT_BDF2, check  # noqa: E402

fs = bench.build_solver(0)
import os
fs.step(np.zeros(2))
fs.step(np.zeros(2))
dev = fs.th.device()
dev.set_dag(True)
n = dev.dag_info(SLOT_BDF2)["tasks"]
st = np.zeros((n, 8), dtype=np.int64)
stage = np.zeros(n, dtype=np.int32)
kind = np.zeros(n, dtype=np.int32)
check(dev.lib.fc_debug_trace_apply(dev._h, SLOT_BDF2, n, st, stage, kind))
t0 = st[:, 0].min()
us = (st[:, :4] - t0) / 100.0
print(f"{n} tasks, apply {us[:, 3].max():.1f} us (first task start to last drain)")
print("stage kind tasks | start first..last | deps met first/median/last | drained first/median/last | wait med | products med | drain med")
for s in range(stage.max() + 1):
    m = stage == s
    u = us[m]
    print(f"{s:3d} {'up' if kind[m][0] == 0 else 'dn'} {m.sum():6d} | {u[:, 0].min():7.2f} {u[:, 0].max():7.2f} | "
          f"{u[:, 1].min():7.2f} {np.median(u[:, 1]):7.2f} {u[:, 1].max():7.2f} | {u[:, 3].min():7.2f} {np.median(u[:, 3]):7.2f} {u[:, 3].max():7.2f} | "
          f"{np.median(u[:, 1] - u[:, 0]):6.2f} | {np.median(u[:, 2] - u[:, 1]):6.2f} | {np.median(u[:, 3] - u[:, 2]):6.2f}")
if len(sys.argv) > 1:
    np.savez_compressed(sys.argv[1], stamps=st, stage=stage, kind=kind)
