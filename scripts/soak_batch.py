"""Soak of the batched step (split tiles, tickets, speculated gather, side-stream tail): N batched steps at k = 32 with identical scenarios
in columns j and j + 16 -- the residual monitor must stay at round-off on every step and the twin columns bit-identical to the end.
    python scripts/soak_batch.py [steps=20000] [k=32]"""
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from flowcontrol_amd.batch import BatchedFlowSolver  # noqa: E402
from flowcontrol_amd.examples.cylinder.cylinderflowsolver import CylinderFlowSolver  # noqa: E402
from flowcontrol_amd.fem.spaces import Function  # noqa: E402
from flowcontrol_amd.flowsolverparameters import ParamIC  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 32
fs = CylinderFlowSolver.make_default(Re=100, path_out=tempfile.mkdtemp(prefix="fc_soak_"), num_steps=0, save_every=0)
U0, P0 = Function(fs.W, np.load(ROOT / "tests" / "golden" / "cylinder_O1.npz")["UP0"]).split()
fs._assign_steady_state(U0, P0)
half = k // 2
ics = [ParamIC(xloc=2.0 + 0.05 * (j % half), yloc=0.02 * (j % half), radius=0.5, amplitude=1.0) for j in range(k)]
bfs = BatchedFlowSolver(fs, k)
bfs.initialize_time_stepping(ics=ics)
worst = 0.0
t0 = time.perf_counter()
for n in range(steps):
    u = np.array([[0.3 * np.sin(0.01 * n + 0.1 * (j % half)), -0.2 * np.cos(0.013 * n)] for j in range(k)])
    y = bfs.step(u)
    assert y is not None and np.all(np.isfinite(y)), n
    if n % 500 == 499:
        worst = max(worst, float(bfs.solve_info[:, 1].max()))
        assert np.array_equal(y[:half], y[half : 2 * half]), f"twin columns differ at step {n}"
        print(f"step {n + 1}: residual max so far {worst:.2e}, {(n + 1) * k / (time.perf_counter() - t0):.0f} sim-steps/s", flush=True)
assert worst < 1e-12
print("OK", steps, "steps, k =", k, "residual max", worst)
