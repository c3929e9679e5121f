"""Where the host's time goes inside BatchedFlowSolver.step (wall clock of the library calls and of the Python between them):
    python scripts/host_split_batch.py [k=16] [steps=400]"""
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

k = int(sys.argv[1]) if len(sys.argv) > 1 else 16
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
fs = CylinderFlowSolver.make_default(Re=100, path_out=tempfile.mkdtemp(), num_steps=10)
g = np.load(ROOT / "tests" / "golden" / "cylinder_O1.npz")
U0, P0 = Function(fs.W, g["UP0"]).split()
fs._assign_steady_state(U0, P0)
fs.params_ic = ParamIC(xloc=2.0, yloc=0.0, radius=0.5, amplitude=1.0)
fs.initialize_time_stepping(ic=None)
fs.step([0.0, 0.0])
bfs = BatchedFlowSolver(fs, k)
bfs.initialize_time_stepping(ics=[fs.params_ic] * k)
u = np.zeros((k, 2))
for _ in range(30):
    bfs.step(u)
dev = bfs.dev
acc = {"begin": 0.0, "collect": 0.0, "end": 0.0}


def wrap(fn, key):
    def inner(*a, **kw):
        t = time.perf_counter()
        r = fn(*a, **kw)
        acc[key] += time.perf_counter() - t
        return r

    return inner


dev.step_batch_begin, dev.step_batch_collect, dev.step_batch_end_early = (wrap(dev.step_batch_begin, "begin"), wrap(dev.step_batch_collect, "collect"),
                                                                          wrap(dev.step_batch_end_early, "end"))
t0 = time.perf_counter()
for _ in range(steps):
    bfs.step(u)
tot = time.perf_counter() - t0
other = tot - sum(acc.values())
print(f"k={k}: {k * steps / tot:.0f} sim-steps/s; per batched step [us]: total {1e6 * tot / steps:.1f} = step_batch_begin {1e6 * acc['begin'] / steps:.1f} + collect "
      f"{1e6 * acc['collect'] / steps:.1f} + end_early {1e6 * acc['end'] / steps:.1f} + python {1e6 * other / steps:.1f}")
bfs.close()
