"""Flow past a cylinder on an MI355X: base flow, closed loop with the shipped LTI controller, restart.

Same user-visible sequence as the reference's ``src/examples/cylinder/run_cylinder_example.py``
(base flow by Picard then Newton, feedback of the first probe to both actuators, checkpoints every
25 steps, a second solver restarting from one of them).

    python -m flowcontrol_amd.examples.cylinder.run_cylinder_example [num_steps]
"""

import logging
import sys
import time
from pathlib import Path


from flowcontrol_amd import utils as flu
from flowcontrol_amd.controller import Controller
from flowcontrol_amd.examples.cylinder.cylinderflowsolver import CylinderFlowSolver
from flowcontrol_amd.flowsolverparameters import ParamIC

CONTROLLER = Path(__file__).resolve().parent / "data_input" / "Kopt_reduced13.mat"
DT = 0.005
SAVE_EVERY = 25


def feedback_loop(fs, K, n):
    """y_1 → K → the same command on both poles, n times; returns the wall-clock seconds."""
    t0 = time.perf_counter()
    for _ in range(n):
        y = flu.MpiUtils.mpi_broadcast(fs.y_meas)
        command = K.step(y=-y[0], dt=fs.params_time.dt)
        fs.step(u_ctrl=[command[0], command[0]])
    return time.perf_counter() - t0


def main(num_steps: int = 1000, path_out: Path | None = None):
    logging.basicConfig(level=logging.INFO)
    out = Path(path_out) if path_out else Path.cwd() / "data_output"

    fs = CylinderFlowSolver.make_default(Re=100, path_out=out, num_steps=num_steps, save_every=SAVE_EVERY, verbose=100)
    fs.params_ic = ParamIC(xloc=2.0, yloc=0.0, radius=0.5, amplitude=1.0)
    t0 = time.perf_counter()
    fs.compute_steady_state(method="picard", max_iter=3, tol=1e-7, u_ctrl=[0.0, 0.0])
    fs.compute_steady_state(method="newton", max_iter=25, u_ctrl=[0.0, 0.0], initial_guess=fs.fields.UP0)
    print(f"base flow: {time.perf_counter() - t0:.2f} s (every iteration assembled, factorised and solved on the GPU)")
    fs.initialize_time_stepping(ic=None)
    K = Controller.from_file(file=CONTROLLER, x0=None)
    setup = feedback_loop(fs, K, 1)  # the first step assembles and factorises the two time-stepping operators
    seconds = feedback_loop(fs, K, num_steps - 1)
    print(f"first step (operators, factorisations) {setup:.2f} s; {num_steps - 1} closed-loop steps: {seconds:.2f} s = "
          f"{(num_steps - 1) / seconds:.0f} steps/s (checkpoint every {SAVE_EVERY} steps)")
    fs.write_timeseries()
    flu.summarize_timings(fs)

    # a second solver picks the run up again at the first checkpoint
    again = CylinderFlowSolver.make_default(Re=100, path_out=out, num_steps=10, save_every=5, Tstart=SAVE_EVERY * DT, verbose=5)
    again.load_steady_state()
    again.initialize_time_stepping(Tstart=again.params_time.Tstart)
    feedback_loop(again, K, again.params_time.num_steps)
    again.write_timeseries()
    print(f"restart from t = {SAVE_EVERY * DT}: y = {again.y_meas}")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 1000)
