"""N time steps of the cylinder case in the factorisation-free Krylov mode (profiling target of scripts/profile_krylov_free.sh).

    python scripts/krylov_free_steps.py [steps=100] [method=gmres]
"""
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from flowcontrol_amd.examples.cylinder.cylinderflowsolver import CylinderFlowSolver  # noqa: E402
from flowcontrol_amd.fem.spaces import Function  # noqa: E402
from flowcontrol_amd.flowsolverparameters import ParamIC  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
method = sys.argv[2] if len(sys.argv) > 2 else "gmres"
fs = CylinderFlowSolver.make_default(Re=100, path_out=tempfile.mkdtemp(prefix="fc_kf_"), num_steps=0, save_every=0)
fs.params_ic = ParamIC(xloc=2.0, yloc=0.0, radius=0.5, amplitude=1.0)
fs.krylov_precond, fs.krylov_method, fs.krylov_max_iter, fs.krylov_rtol = "schur_amg", method, 300, 1e-10
U0, P0 = Function(fs.W, np.load(ROOT / "tests" / "golden" / "cylinder_O1.npz")["UP0"]).split()
fs._assign_steady_state(U0, P0)
fs.initialize_time_stepping(ic=None)
u0 = np.zeros(2)
for _ in range(3):
    fs.step(u0)
its = []
t0 = time.perf_counter()
for _ in range(steps):
    fs.step(u0)
    its.append(int(fs.solve_info[0]))
dt = time.perf_counter() - t0
print(f"{method}: {steps} steps, {np.mean(its):.2f} iterations/step (max {max(its)}), {steps / dt:.1f} steps/s, residual {fs.solve_info[1]:.2e}, "
      f"held {fs.th.device().krylov_info(1)['bytes'] / 1e6:.1f} MB")
fs.th.release_device()
