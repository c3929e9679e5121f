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
us = (st[:, :5] - t0) / 100.0
print(f"{n} tasks, apply {us[:, 4].max():.1f} us (first entry to last drain)")
print("stage kind tasks | entry first..last | deps met first/median/last | done first/median/last | wait med | products med | drain med")
for s in range(stage.max() + 1):
    m = stage == s
    u = us[m]
    print(f"{s:3d} {'up' if kind[m][0] == 0 else 'dn'} {m.sum():6d} | {u[:, 0].min():7.2f} {u[:, 0].max():7.2f} | "
          f"{u[:, 2].min():7.2f} {np.median(u[:, 2]):7.2f} {u[:, 2].max():7.2f} | {u[:, 4].min():7.2f} {np.median(u[:, 4]):7.2f} {u[:, 4].max():7.2f} | "
          f"{np.median(u[:, 2] - u[:, 1]):6.2f} | {np.median(u[:, 3] - u[:, 2]):6.2f} | {np.median(u[:, 4] - u[:, 3]):6.2f} | prefetch issue {np.median(u[:, 1] - u[:, 0]):5.2f}")
if len(sys.argv) > 1:
    np.savez_compressed(sys.argv[1], stamps=st, stage=stage, kind=kind)
