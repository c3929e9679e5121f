"""Experiment: (a) cost of the parts of the batched tail (energy on / off; FC_BATCH_DEBUG_NORES=1 in the environment skips
the residual loop), (b) do two handles (two HIP streams) stepping k simulations each overlap on one GPU?"""
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from flowcontrol_amd._lib import SLOT_BDF2  # noqa: E402
from flowcontrol_amd.batch import BatchedFlowSolver  # noqa: E402
from flowcontrol_amd.examples.cylinder.cylinderflowsolver import CylinderFlowSolver  # noqa: E402
from flowcontrol_amd.fem.spaces import Function  # noqa: E402
from flowcontrol_amd.fem.spaces import TaylorHood  # noqa: E402
from flowcontrol_amd.flowsolverparameters import ParamIC  # noqa: E402

g = np.load(ROOT / "tests" / "golden" / "cylinder_O1.npz")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
NH = int(sys.argv[2]) if len(sys.argv) > 2 else 2
STEPS = 300


def make():
    fs = CylinderFlowSolver.make_default(Re=100, path_out=tempfile.mkdtemp(), num_steps=10)
    U0, P0 = Function(fs.W, g["UP0"]).split()
    fs._assign_steady_state(U0, P0)
    fs.params_ic = ParamIC(xloc=2.0, yloc=0.0, radius=0.5, amplitude=1.0)
    fs.initialize_time_stepping(ic=None)
    b = BatchedFlowSolver(fs, K)
    import os
    if os.environ.get('FC_BATCH_DEBUG_TAIL'):
        b.residual_tol = np.inf
    b.initialize_time_stepping(ics=[fs.params_ic] * K)
    for _ in range(3):
        b.step(np.zeros((K, 2)))
    return fs, b


solvers = [make() for _ in range(NH)]
u = np.zeros((K, 2))
for energy in (True, False):
    dev = solvers[0][1].dev
    for _ in range(20):
        dev.step_batch(SLOT_BDF2, u, compute_energy=energy)
    t0 = time.perf_counter()
    for _ in range(STEPS):
        dev.step_batch(SLOT_BDF2, u, compute_energy=energy)
    dt = (time.perf_counter() - t0) / STEPS
    print(f"one handle, k={K}, energy={energy}: {dt * 1e6:.1f} us per batched step, {K / dt:.0f} sim-steps/s", flush=True)
devs = [b.dev for _, b in solvers]
for _ in range(20):
    for d in devs:
        d.step_batch_begin(SLOT_BDF2, u, True)
    for d in devs:
        d.step_batch_end()
t0 = time.perf_counter()
for _ in range(STEPS):
    for d in devs:
        d.step_batch_begin(SLOT_BDF2, u, True)
    for d in devs:
        d.step_batch_end()
dt = (time.perf_counter() - t0) / STEPS
print(f"{NH} handles x k={K} interleaved (begin all, end all): {dt * 1e6:.1f} us per round, {NH * K / dt:.0f} sim-steps/s", flush=True)
