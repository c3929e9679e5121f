"""A/B of the factor apply on the headline workload: one launch (fc_nd_dag) vs one launch per tree level.
    python scripts/ab_dag.py [steps]"""
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402
from flowcontrol_amd._lib import SLOT_BDF2  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
bench.REFINE = int(sys.argv[2]) if len(sys.argv) > 2 else 0
fs = bench.build_solver(0)
u0 = np.zeros(2)
fs.step(u0)
dev = fs.th.device()
out = {"dag_info": dev.dag_info(SLOT_BDF2)}
for mode in ("dag", "levels", "dag", "levels"):
    dev.set_dag(mode == "dag")
    for _ in range(50):
        fs.step(u0)
    t0 = time.perf_counter()
    for _ in range(steps):
        fs.step(u0)
    dt = time.perf_counter() - t0
    ms, nl = dev.bench_sweeps(SLOT_BDF2, 300)
    phases, _ = dev.profile_steps(SLOT_BDF2, 50, u0)
    out.setdefault(mode, []).append({"steps_per_s": steps / dt, "apply_us": ms * 1e3, "launches": nl, "phases_us": [float(1e3 * p) for p in phases]})
out["dag_info_end"] = dev.dag_info(SLOT_BDF2)
sweep_bytes, _ = dev.algorithmic_bytes(SLOT_BDF2)
out["sweep_bytes"] = sweep_bytes
print(json.dumps(out, indent=1))
