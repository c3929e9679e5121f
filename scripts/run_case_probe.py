"""Throughput probes of the larger BASELINE cases on one MI355X:
    python scripts/run_case_probe.py cavity_fine    # config 3 ingredients: open cavity Re = 7500 on the reference's
                                                    # cavity_fine mesh (193 916 cells, 876 645 dofs), FORCE actuator
    python scripts/run_case_probe.py pinball        # config 5 ingredients: fluidic pinball Re = 100, 3 BC actuators
Prints set-up and stepping rates (the base flow is only a few Picard sweeps: a throughput probe, not a
converged base flow)."""
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: F401,E402  (one HIP runtime, loaded first)

case = sys.argv[1] if len(sys.argv) > 1 else "cavity_fine"
t0 = time.time()
if case == "pinball":
    from flowcontrol_amd.actuator import CYLINDER_ACTUATION_MODE  # noqa: E402
    from flowcontrol_amd.examples.pinball.pinballflowsolver import PinballFlowSolver  # noqa: E402

    fs = PinballFlowSolver.make_default(Re=100, mode_actuation=CYLINDER_ACTUATION_MODE.SUCTION, path_out=tempfile.mkdtemp(), num_steps=100)
else:
    from flowcontrol_amd.examples.cavity.cavityflowsolver import CavityFlowSolver  # noqa: E402

    fs = CavityFlowSolver.make_default(Re=7500, path_out=tempfile.mkdtemp(), num_steps=100,
                                       meshpath=ROOT / "tests" / "golden" / "meshes" / "cavity_fine.npz")
n_act = len(fs.params_control.actuator_list)
print(f"solver object {time.time() - t0:.1f} s; N = {fs.th.N}", flush=True)
t0 = time.time()
fs.compute_steady_state(method="picard", max_iter=2, tol=1e-7, u_ctrl=[0.0] * n_act)
print(f"2 Picard sweeps on the device {time.time() - t0:.1f} s", flush=True)
fs.initialize_time_stepping(ic=None)
t0 = time.time()
fs.step([0.0] * n_act)
print(f"first step (2 operators: structure + device factorisation) {time.time() - t0:.1f} s; "
      f"refactor {fs.th.device().refactor_ms} ms; factor nnz {fs.th.device().factor_nnz}", flush=True)
for _ in range(10):
    fs.step([0.0] * n_act)
n = 200
t0 = time.time()
for k in range(n):
    y = fs.step([0.01 * np.sin(0.05 * k)] * n_act)
dt = (time.time() - t0) / n
print(f"synchronous: {1 / dt:.1f} steps/s ({1e3 * dt:.3f} ms/step); residual {fs.solve_info[1]:.2e}; y {y}", flush=True)
t0 = time.time()
fs.run(n, np.zeros(n_act))
dt = (time.time() - t0) / n
dev = fs.th.device()
from flowcontrol_amd._lib import SLOT_BDF2  # noqa: E402
sweep_bytes, _ = dev.algorithmic_bytes(SLOT_BDF2)
print(f"batched: {1 / dt:.1f} steps/s; factor sweep bytes/step {sweep_bytes / 1e9:.2f} GB -> >= {sweep_bytes / dt / 1e12:.2f} TB/s "
      f"if the sweeps were the whole step", flush=True)
