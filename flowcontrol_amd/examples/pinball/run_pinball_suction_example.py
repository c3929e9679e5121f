"""Fluidic pinball (three cylinders, Re = 100), suction / blowing actuation, on an MI355X.

Same user-visible sequence as the reference's ``src/examples/pinball/run_pinball_suction_example.py``: three
``ActuatorBCParabolicV`` slots of 10° on the cylinders, base flow by Picard ×15 from the antisymmetric_bot guess then
Newton ×10, open-loop Gaussian bumps on the three slots (peaks +2.0 / −1.5 / −2.0 at t = 0.25 / 0.5 / 0.75, width 0.1),
force coefficients of the final state.

    python -m flowcontrol_amd.examples.pinball.run_pinball_suction_example [num_steps]
"""
import logging
import sys
import time
from pathlib import Path

from flowcontrol_amd import utils as flu
from flowcontrol_amd.actuator import CYLINDER_ACTUATION_MODE
from flowcontrol_amd.examples.pinball.pinballflowsolver import PinballCustomInitialGuess, PinballFlowSolver
from flowcontrol_amd.examples.pinball.run_pinball_rotation_example import gaussian_bump
from flowcontrol_amd.flowsolverparameters import ParamIC

PEAK_TIMES, PEAK_VALUES = (0.25, 0.5, 0.75), (+2.0, -1.5, -2.0)


def main(num_steps: int = 20, path_out: Path | None = None):
    logging.basicConfig(level=logging.INFO)
    out = Path(path_out) if path_out else Path.cwd() / "data_output"
    fs = PinballFlowSolver.make_default(Re=100, mode_actuation=CYLINDER_ACTUATION_MODE.SUCTION, path_out=out, num_steps=num_steps,
                                        save_every=10, verbose=10)
    fs.params_ic = ParamIC(xloc=2.0, yloc=0.0, radius=0.5, amplitude=1.0)
    flu.export_subdomains(fs.mesh, fs.boundaries.subdomain, out / "subdomains.xdmf")
    start = PinballCustomInitialGuess(mode="antisymmetric_bot").as_dolfin_function(function_space=fs.W)
    t0 = time.perf_counter()
    fs.compute_steady_state(method="picard", max_iter=15, tol=1e-7, u_ctrl=[0.0, 0.0, 0.0], initial_guess=start)
    fs.compute_steady_state(method="newton", max_iter=10, u_ctrl=[0.0, 0.0, 0.0], initial_guess=fs.fields.UP0)
    print(f"base flow on {fs.th.N} dofs: {time.perf_counter() - t0:.2f} s")
    fs.initialize_time_stepping(ic=None)
    t0 = time.perf_counter()
    for _ in range(fs.params_time.num_steps):
        fs.step(u_ctrl=[a * gaussian_bump(fs.t, tp) for a, tp in zip(PEAK_VALUES, PEAK_TIMES)])
    seconds = time.perf_counter() - t0
    print(f"{num_steps} steps: {seconds:.2f} s = {num_steps / seconds:.0f} steps/s; y = {fs.y_meas}")
    fs.write_timeseries()
    for surface, (cl, cd) in fs.compute_force_coefficients(fs.fields.u_, fs.fields.p_).items():
        print(f"{surface}: Cl={cl:.4f}, Cd={cd:.4f}")
    return fs


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 20)
