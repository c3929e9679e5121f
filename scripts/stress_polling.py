"""Eager (host polls the zero-copy step record) vs batched (one synchronisation at the end) runs of the
same open-loop trajectory must agree bit for bit: a stale or torn record read would show up here."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: F401  (as bench.py: one HIP runtime, loaded first)
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
u = np.stack([0.05 * np.sin(0.01 * np.arange(n)), -0.02 * np.cos(0.013 * np.arange(n))], axis=1)
fa = bench.build_solver(0)
ya = np.array([fa.step(u[k]).copy() for k in range(n)])
ea = fa.timeseries["dE"].to_numpy()[1:]
fa.th.release_device()
fb = bench.build_solver(0)
fb.step(u[0])
yb, eb = fb.run(n - 1, u[1:])
yb = np.vstack([fb.timeseries[["y_meas_1", "y_meas_2", "y_meas_3"]].to_numpy()[1:2], yb])
print("steps", n, "max |y_eager - y_batched|", np.abs(ya - yb).max(), "rows differing", int(np.any(ya != yb, axis=1).sum()),
      "max |dE diff|", np.nanmax(np.abs(ea[1:] - eb)))
dup = int(np.all(ya[1:] == ya[:-1], axis=1).sum())
print("consecutive identical eager rows:", dup)
sys.exit(0 if (np.array_equal(ya, yb) and dup == 0) else 1)
