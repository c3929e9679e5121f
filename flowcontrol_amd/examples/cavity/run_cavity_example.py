"""Flow over an open cavity (Re = 7500) on an MI355X: base flow, then a time simulation with the
Gaussian volume-force actuator and the wall-shear sensor.

Same user-visible sequence as the reference's ``src/examples/cavity/run_cavity_example.py`` (Picard ×10 →
Newton ×10, IC amplitude 0.1 at (2, 0), unactuated steps, checkpoints every 50 steps); ``--closed-loop`` feeds the
wall-shear fluctuation through a first-order low-pass ``Controller`` into the actuator (BASELINE config 3; no
cavity controller file ships with the reference).

    python -m flowcontrol_amd.examples.cavity.run_cavity_example [num_steps] [--fine] [--closed-loop]
"""
import logging
import sys
import time
from pathlib import Path

from flowcontrol_amd.controller import Controller
from flowcontrol_amd.examples.cavity.cavityflowsolver import DEFAULT_MESH, CavityFlowSolver
from flowcontrol_amd.flowsolverparameters import ParamIC


def main(num_steps: int = 10000, fine: bool = False, closed_loop: bool = False, path_out: Path | None = None):
    logging.basicConfig(level=logging.INFO)
    out = Path(path_out) if path_out else Path.cwd() / "data_output"
    mesh = DEFAULT_MESH.with_name("cavity_fine.npz") if fine else DEFAULT_MESH
    fs = CavityFlowSolver.make_default(Re=7500, path_out=out, num_steps=num_steps, save_every=50, verbose=10, meshpath=mesh)
    fs.params_ic = ParamIC(xloc=2.0, yloc=0.0, radius=0.5, amplitude=0.1)
    t0 = time.perf_counter()
    fs.compute_steady_state(method="picard", max_iter=10, tol=1e-7, u_ctrl=[0.0])
    fs.compute_steady_state(method="newton", max_iter=10, u_ctrl=[0.0], initial_guess=fs.fields.UP0)
    print(f"base flow on {fs.th.N} dofs: {time.perf_counter() - t0:.2f} s")
    fs.initialize_time_stepping(ic=None)
    K = Controller(A=[[-100.0]], B=[[1.0]], C=[[50.0]], D=[[0.0]]) if closed_loop else None
    y0 = fs.y_meas[0]
    t0 = time.perf_counter()
    for _ in range(fs.params_time.num_steps):
        u = K.step(y=fs.y_meas[0] - y0, dt=fs.params_time.dt)[0] if K else 0.0
        fs.step(u_ctrl=[u])
    dt = time.perf_counter() - t0
    print(f"{num_steps} steps: {dt:.2f} s = {num_steps / dt:.0f} steps/s; y = {fs.y_meas}")
    fs.write_timeseries()


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    main(int(args[0]) if args else 10000, fine="--fine" in sys.argv, closed_loop="--closed-loop" in sys.argv)
