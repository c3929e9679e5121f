"""1000-step actuated open-loop series of the cylinder Re=100 case from the CPU oracle (north_star:
"sensor timeseries within 1e-6 rel-L2 of reference over 1000 steps").

    python tests/golden/make_cylinder_1000_steps.py        (≈ 90 s on one core)

Starts from the base flow of tests/golden/cylinder_O1.npz (UP0) and the IC of
run_cylinder_example.py:55 (ParamIC(xloc=2, yloc=0, radius=0.5, amplitude=1)); the actuation is the
deterministic schedule u_ctrl[k] = (0.05 sin(0.01 k), -0.02 cos(0.013 k)) so that the Dirichlet lifting
is exercised at every step.  Output tests/golden/cylinder_O1_ol1000.npz: y (1001, 3), dE (1001,).
"""
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from flowcontrol_amd.examples.cylinder.cylinderflowsolver import CylinderFlowSolver  # noqa: E402
from oracle import ns_oracle as O  # noqa: E402

HERE = Path(__file__).parent
N_STEPS = 1000


def schedule(n):
    k = np.arange(n)
    return np.stack([0.05 * np.sin(0.01 * k), -0.02 * np.cos(0.013 * k)], axis=1)


def main():
    fs = CylinderFlowSolver.make_default(path_out=tempfile.mkdtemp())
    th = fs.th
    d = O.Disc.from_taylor_hood(th)
    up0 = np.load(HERE / "cylinder_O1.npz")["UP0"]
    U0 = up0[: 2 * th.nn]
    dofs, prof = fs._bc_tables()
    ts = O.TimeStepper(d, fs.params_flow.Re, fs.params_time.dt, U0, dofs, prof)
    M = O.velocity_mass(d)
    rows = [s.row(fs) for s in fs.params_control.sensor_list]
    meas = lambda v: np.array([w @ v[i] for i, w in rows])  # noqa: E731
    uic = O.div0_gaussian_nodal(th.node_coords, 2.0, 0.0, 0.5)
    u_n = np.r_[uic[:, 0], uic[:, 1]]
    u_nn = u_n.copy()
    ys, dEs = [meas(np.r_[u_n, up0[2 * th.nn :]])], [0.5 * u_n @ (M @ u_n)]
    u = schedule(N_STEPS)
    order, t0 = 1, time.time()
    for k in range(N_STEPS):
        upn = ts.step(order, u_n, u_nn, u[k])
        order = 2
        u_nn, u_n = u_n, upn[: 2 * th.nn]
        ys.append(meas(upn)), dEs.append(0.5 * u_n @ (M @ u_n))
        if (k + 1) % 100 == 0:
            print(f"step {k + 1}: y {ys[-1]} dE {dEs[-1]:.6g}  ({time.time() - t0:.0f} s)", flush=True)
    np.savez_compressed(HERE / "cylinder_O1_ol1000.npz", y=np.array(ys), dE=np.array(dEs))
    print("wrote", HERE / "cylinder_O1_ol1000.npz")


if __name__ == "__main__":
    main()
