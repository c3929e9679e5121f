import os, sys, tempfile
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
mode = sys.argv[1]
os.environ["FC_PY_SYMBOLIC"] = mode
from flowcontrol_amd._lib import SLOT_BDF2, SLOT_BDF1
from flowcontrol_amd.examples.cylinder.cylinderflowsolver import CylinderFlowSolver
from flowcontrol_amd.fem.spaces import Function
from flowcontrol_amd.flowsolverparameters import ParamIC
g = np.load(ROOT / "tests/golden/cylinder_O1.npz")
for method in ("gmres", "bicgstab"):
    fs = CylinderFlowSolver.make_default(Re=100, path_out=tempfile.mkdtemp(), num_steps=6)
    fs.params_ic = ParamIC(xloc=2.0, yloc=0.0, radius=0.5, amplitude=1.0)
    fs.nd_truncate, fs.krylov_method, fs.krylov_max_iter = 1, method, 800
    U0, P0 = Function(fs.W, g["UP0"]).split()
    fs._assign_steady_state(U0, P0)
    fs.initialize_time_stepping(ic=None)
    dev = fs.th.device()
    try:
        fs.step([0.0, 0.0])
        print("  first step", fs.solve_info)
    except Exception as e:
        print("  first step ERR", e)
    fv = dev.factor_values(SLOT_BDF1)
    print(mode, method, "nvals", fv.size, "finite", np.isfinite(fv).all(), "sum", fv.sum(), "abs", np.abs(fv).sum())
    b = np.cos(0.37 * np.arange(dev.N) + 0.1)
    for slot in (SLOT_BDF1, SLOT_BDF2):
        try:
            x, info = dev.solve(slot, b)
            print("  solve slot", slot, info, "x sum", x.sum())
        except Exception as e:
            print("  solve slot", slot, "ERR", e)
    for k in range(3):
        try:
            fs.step([0.0, 0.0])
            print("  step", k, fs.solve_info)
        except Exception as e:
            print("  step", k, "ERR", e)
            break
    fs.th.release_device()
